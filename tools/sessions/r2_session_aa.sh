#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_aa
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_all.txt 2>&1
grep -E "passed|failed|Error" $OUT/pytest_all.txt | tail -3
for i in 1 2 3 4; do
python3 bench.py --no-cpu > $OUT/bench_search_$i.json 2> $OUT/bench_search_$i.err
done
python3 bench.py --no-cpu --size 200 --steps 1 > $OUT/bench_n200.json 2> $OUT/bench_n200.err
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d=json.load(open(f)); c=d["config"]; r=d["roofline"]
        print(f.split("/")[-1], d["value"], "ms/step %.1f"%d["ms_per_step"], "spmv %.4f frac %.3f"%(r["avg_launch_ms"], r["frac"]), c.get("placement_search"))
    except Exception as e: print(f, "ERR", e)
PY
