#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_c
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_all.txt 2>&1
tail -5 $OUT/pytest_all.txt
export STAN_HIP_LIB=$R/stan_amd/csrc/build_lab/libstan_hip_lab.so
timeout 600 python3 tools/placement_alloc.py 148 > $OUT/placement_alloc_n148.txt 2>&1
unset STAN_HIP_LIB
cat $OUT/placement_alloc_n148.txt
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 900 python3 tools/cpu_sizes.py 100 > $OUT/cpu_sizes_n100.jsonl 2> $OUT/cpu_sizes.err
cat $OUT/cpu_sizes_n100.jsonl | cut -c1-600
