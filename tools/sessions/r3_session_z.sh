#!/bin/bash
# round 3, session z: fuzz sweep with folded rows forced on; full suite on the final build; final default bench + profile
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_z
mkdir -p $OUT
cd $R
timeout 1500 python3 tools/fuzz_parity.py 0 300 1 > $OUT/fuzz_sweep_folded_0_300.txt 2>&1
echo "fuzz rc=$?"; tail -4 $OUT/fuzz_sweep_folded_0_300.txt | cut -c1-300
timeout 3000 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; grep -n "passed\|failed" $OUT/pytest_gpu.txt | tail -3
