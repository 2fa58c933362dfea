#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_s
mkdir -p $OUT
cd $R
timeout 1500 python3 bench.py --size 400 --steps 1 --warmup 0 --no-cpu > $OUT/bench_n400_fp64.json 2> $OUT/bench_n400_fp64.err
timeout 1500 python3 bench.py --size 400 --steps 1 --warmup 0 --no-cpu --fixed48 > $OUT/bench_n400_fixed48.json 2> $OUT/bench_n400_fixed48.err
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d=json.load(open(f)); c=d["config"]; r=d["roofline"]
        print(f.split("/")[-1], d["value"], "ms/step %.1f"%d["ms_per_step"], "spmv %.4f frac %.3f"%(r["avg_launch_ms"], r["frac"]), "its", c["cg_iterations"], c["matrix_format"], c.get("placement_search"))
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-500:])
PY
