#!/bin/bash
# round 3, session ai: GPU suite + smoke on the final tree
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_ai
mkdir -p $OUT
cd $R
python3 __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.txt
timeout 3000 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; grep -n "passed\|failed" $OUT/pytest_gpu.txt | tail -2
