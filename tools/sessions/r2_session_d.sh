#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_d
mkdir -p $OUT
cd $R
rocm-smi --showclocks --showpower --showtemp > $OUT/smi_before.txt 2>&1
export STAN_HIP_LIB=$R/stan_amd/csrc/build_lab/libstan_hip_lab.so
timeout 600 python3 tools/placement_rounds.py 148 6 40 5 0 > $OUT/placement_rounds_n148.txt 2>&1
timeout 600 python3 tools/placement_rounds.py 148 6 20 5 200 > $OUT/placement_rounds_n148_pause200.txt 2>&1
unset STAN_HIP_LIB
rocm-smi --showclocks --showpower --showtemp > $OUT/smi_after.txt 2>&1
timeout 600 python3 -m pytest tests/test_gpu_direct.py tests/test_gpu_round2.py -m gpu -x -q > $OUT/pytest_direct.txt 2>&1
tail -3 $OUT/pytest_direct.txt
cat $OUT/placement_rounds_n148.txt
tail -12 $OUT/placement_rounds_n148_pause200.txt
