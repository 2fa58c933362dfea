#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_i
mkdir -p $OUT
cd $R
bash tools/lib_lab.sh tools/cg_time.py "-DSTAN_P_NT=0 -DSTAN_R_NT=0" "-DSTAN_P_NT=1 -DSTAN_R_NT=0" "-DSTAN_P_NT=1 -DSTAN_R_NT=1" "-DSTAN_P_NT=0 -DSTAN_R_NT=0" "-DSTAN_P_NT=1 -DSTAN_R_NT=1" "-DSTAN_P_NT=1 -DSTAN_R_NT=1 -DSTAN_VEC_NT=2" > $OUT/nt_stores_ab.txt 2>&1
cat $OUT/nt_stores_ab.txt
python3 bench.py --no-cpu > $OUT/bench_default.json 2> $OUT/bench.err
python3 bench.py --no-cpu --placement-tries 1 > $OUT/bench_plain.json 2>> $OUT/bench.err
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    d=json.load(open(f)); c=d["config"]; r=d["roofline"]
    print(f.split("/")[-1], d["value"], "ms/step %.1f"%d["ms_per_step"], "spmv %.4f frac %.3f"%(r["avg_launch_ms"], r["frac"]), c.get("placement_search"))
PY
