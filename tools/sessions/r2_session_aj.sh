#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_aj
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -m gpu -q > $OUT/pytest_all.txt 2>&1
grep -E "passed|failed|Error" $OUT/pytest_all.txt | tail -3
grep -E "^E " $OUT/pytest_all.txt | head -20
python3 tools/small_sizes.py 8,16,24,32,40,50 > $OUT/small_sizes.txt 2>&1
cat $OUT/small_sizes.txt
