#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_y
mkdir -p $OUT
cd $R
python3 bench.py --steps 20 --warmup 1 --no-cpu > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
python3 -c "
import json; d=json.load(open('$OUT/bench_steps20.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['placement_search'])"
