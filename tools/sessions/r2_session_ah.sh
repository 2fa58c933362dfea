#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_ah
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_all.txt 2>&1
grep -E "passed|failed|Error" $OUT/pytest_all.txt | tail -3
tail -30 $OUT/pytest_all.txt | grep -E "^E|assert" | head -20
python3 tools/small_sizes.py 8,16,24,40,56,72 > $OUT/small_sizes.txt 2>&1
cat $OUT/small_sizes.txt
