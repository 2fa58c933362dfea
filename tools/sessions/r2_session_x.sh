#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_x
mkdir -p $OUT
cd $R
timeout 1500 python3 tools/packed_ab.py 148,200 fixed48,fp64 2 > $OUT/packed_ab_final.txt 2>&1
cat $OUT/packed_ab_final.txt
