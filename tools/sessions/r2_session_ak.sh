#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_ak
mkdir -p $OUT
cd $R
python3 tools/small_sizes.py 40,50,56,64,72,80,100 > $OUT/small_sizes_crossover.txt 2>&1
cat $OUT/small_sizes_crossover.txt
