#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_ab
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_all.txt 2>&1
grep -E "passed|failed|Error" $OUT/pytest_all.txt | tail -5
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu > $OUT/bench_under_rocprofv3.json 2> $OUT/bench_under_rocprofv3.err
cd $R
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_summary.py $F > $OUT/kernel_trace_summary.txt 2>&1
S=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/kernel_stats.csv
rm -rf $OUT/trace
bash tools/pmc_run.sh gpurun_out/r02_ab/pmc > $OUT/pmc_summary_stdout.txt 2>&1
find $OUT/pmc -name "*.csv" -delete
python3 bench.py --no-cpu --fixed48 > $OUT/bench_fixed48.json 2> $OUT/bench_fixed48.err
python3 bench.py --no-cpu --mixed > $OUT/bench_mixed.json 2> $OUT/bench_mixed.err
python3 bench.py --no-cpu --size 200 --steps 1 > $OUT/bench_n200.json 2> $OUT/bench_n200.err
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d=json.load(open(f)); c=d["config"]; r=d["roofline"]
        print(f.split("/")[-1], d["value"], "ms/step %.1f"%d["ms_per_step"], "spmv %.4f frac %.3f"%(r["avg_launch_ms"], r["frac"]), c.get("placement_search"))
    except Exception as e: print(f, "ERR", e)
PY
head -8 $OUT/kernel_trace_summary.txt; head -4 $OUT/pmc/pmc_summary.txt
