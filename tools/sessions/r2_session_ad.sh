#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_ae
mkdir -p $OUT
cd $R
export STAN_HIP_LIB=$R/stan_amd/csrc/build_lab/libstan_hip_lab.so
for i in 1 2 3; do timeout 300 python3 tools/placement_vecalloc.py 148 >> $OUT/placement_vecalloc.txt 2>&1; done
grep -E "SpMV|vectors in" $OUT/placement_vecalloc.txt
