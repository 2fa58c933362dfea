#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_u
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_all.txt 2>&1
grep -E "passed|failed|Error" $OUT/pytest_all.txt | tail -3
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench.err
python3 -c "
import json; d=json.load(open('$OUT/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['placement_search'])"
python3 tools/cli_scale.py 100 > $OUT/cli_scale_100.txt 2>&1; tail -5 $OUT/cli_scale_100.txt
