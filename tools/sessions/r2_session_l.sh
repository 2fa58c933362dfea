#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_l
mkdir -p $OUT
cd $R
export STAN_HIP_LIB=$R/stan_amd/csrc/build_lab/libstan_hip_lab.so
timeout 900 python3 tools/fold_ab.py 148 1 > $OUT/fold_ab_n148.txt 2>&1
unset STAN_HIP_LIB
tail -3 $OUT/fold_ab_n148.txt
timeout 1500 python3 tools/fuzz_parity.py 0 400 > $OUT/fuzz_sweep_0_400.txt 2>&1
tail -3 $OUT/fuzz_sweep_0_400.txt
bash tools/shard_loop.sh 6 fuzz:124 2 > $OUT/shard_loop_6.txt 2>&1
bash tools/shard_loop.sh 7 fuzz:101 2 > $OUT/shard_loop_7.txt 2>&1
cat $OUT/shard_loop_6.txt $OUT/shard_loop_7.txt
