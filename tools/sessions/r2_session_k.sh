#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_k
mkdir -p $OUT
cd $R
timeout 1200 python3 -m pytest tests/test_gpu_multi.py tests/test_gpu_sharded.py -m gpu -x -q > $OUT/pytest_multi.txt 2>&1
tail -30 $OUT/pytest_multi.txt
