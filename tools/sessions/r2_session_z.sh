#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_z
mkdir -p $OUT
cd $R
export STAN_HIP_LIB=$R/stan_amd/csrc/build_lab/libstan_hip_lab.so
timeout 600 python3 tools/placement_cross.py 148 12 > $OUT/placement_cross_self_n148.txt 2>&1
cat $OUT/placement_cross_self_n148.txt
