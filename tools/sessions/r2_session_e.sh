#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_e
mkdir -p $OUT
cd $R
for i in 1 2 3; do
  python3 bench.py --no-cpu > $OUT/bench_search_$i.json 2> $OUT/bench_search_$i.err
done
python3 bench.py --no-cpu --placement-tries 1 > $OUT/bench_plain_alloc.json 2> $OUT/bench_plain.err
python3 bench.py --size 100 --mixed --etype 1 --no-cpu --steps 1 --warmup 0 > $OUT/bench_n100_mixed_g1.json 2> $OUT/bench_n100_mixed_g1.err
python3 bench.py --size 200 --no-cpu --steps 1 --warmup 1 > $OUT/bench_n200_fp64.json 2> $OUT/bench_n200.err
timeout 1500 python3 bench.py --size 400 --mixed --etype 1 --max-its 300 --steps 1 --warmup 0 --no-cpu --placement-tries 1 > $OUT/bench_n400_mixed_g1_capped300.json 2> $OUT/bench_n400.err
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d=json.load(open(f)); c=d["config"]; r=d["roofline"]
        print(f.split("/")[-1], d["value"], "ms/step %.1f"%d["ms_per_step"], "spmv %.4f frac %.3f"%(r["avg_launch_ms"], r["frac"]), "its", c["cg_iterations"], "type", c["termination_type"], "rel %.2e"%c["rel_residual"], c.get("placement_search"))
    except Exception as e: print(f, "ERR", e)
PY
