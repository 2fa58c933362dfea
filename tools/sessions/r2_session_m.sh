#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_w
mkdir -p $OUT
cd $R
export STAN_HIP_LIB=$R/stan_amd/csrc/build_lab/libstan_hip_lab.so
timeout 900 python3 tools/fold_ab.py 148 1 > $OUT/fold_ab_n148.txt 2>&1
tail -2 $OUT/fold_ab_n148.txt
