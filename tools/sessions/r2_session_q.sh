#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_q
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_all.txt 2>&1
grep -E "passed|failed|Error" $OUT/pytest_all.txt | tail -5
timeout 900 python3 tools/fold_ab.py 148 2 > $OUT/fold_ab_n148.txt 2>&1
grep -E "deferred|placement" $OUT/fold_ab_n148.txt
python3 bench.py --no-cpu > $OUT/bench_default.json 2> $OUT/bench.err
python3 -c "
import json; d=json.load(open('$OUT/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
