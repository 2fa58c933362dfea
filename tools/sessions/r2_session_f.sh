#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_f
mkdir -p $OUT
cd $R
export STAN_HIP_LIB=$R/stan_amd/csrc/build_lab/libstan_hip_lab.so
timeout 600 python3 tools/placement_cross.py 148 16 > $OUT/placement_cross_n148.txt 2>&1
unset STAN_HIP_LIB
cat $OUT/placement_cross_n148.txt
for i in 1 2; do
python3 bench.py --no-cpu --placement-tries 1 > $OUT/bench_plain_$i.json 2> $OUT/bench_plain_$i.err
python3 bench.py --no-cpu > $OUT/bench_search_$i.json 2> $OUT/bench_search_$i.err
done
timeout 900 python3 -m pytest tests/test_gpu_multi.py tests/test_gpu_sharded.py tests/test_gpu_parity.py -m gpu -x -q -k "multi or shard or rank" > $OUT/pytest_shard.txt 2>&1
tail -3 $OUT/pytest_shard.txt
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d=json.load(open(f)); c=d["config"]; r=d["roofline"]
        print(f.split("/")[-1], d["value"], "ms/step %.1f"%d["ms_per_step"], "spmv %.4f frac %.3f"%(r["avg_launch_ms"], r["frac"]), c.get("placement_search"))
    except Exception as e: print(f, "ERR", e)
PY
