#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_g
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_all.txt 2>&1
grep -E "passed|failed|Error" $OUT/pytest_all.txt | tail -5
for i in 1 2 3; do
python3 bench.py --no-cpu > $OUT/bench_search_$i.json 2> $OUT/bench_search_$i.err
python3 bench.py --no-cpu --placement-tries 1 > $OUT/bench_plain_$i.json 2> $OUT/bench_plain_$i.err
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d=json.load(open(f)); c=d["config"]; r=d["roofline"]
        print(f.split("/")[-1], d["value"], "ms/step %.1f"%d["ms_per_step"], "spmv %.4f frac %.3f"%(r["avg_launch_ms"], r["frac"]), c.get("placement_search"))
    except Exception as e: print(f, "ERR", e)
PY
