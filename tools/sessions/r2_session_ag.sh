#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_ag
mkdir -p $OUT
cd $R
python3 tools/small_sizes.py > $OUT/small_sizes.txt 2>&1
cat $OUT/small_sizes.txt
