#!/bin/bash
# round 3, session ad: fuzz sweep through the folded streams with the final kernel (wave-local exchange), default-flag fuzz too
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_ad
mkdir -p $OUT
cd $R
timeout 1500 python3 tools/fuzz_parity.py 300 300 1 > $OUT/fuzz_sweep_folded_300_600.txt 2>&1
echo "fuzz (folded) rc=$?"; tail -3 $OUT/fuzz_sweep_folded_300_600.txt | cut -c1-300
timeout 1500 python3 tools/fuzz_parity.py 300 300 > $OUT/fuzz_sweep_300_600.txt 2>&1
echo "fuzz (default) rc=$?"; tail -2 $OUT/fuzz_sweep_300_600.txt | cut -c1-300
