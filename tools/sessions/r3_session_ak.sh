#!/bin/bash
# round 3, session ak: the irregular mesh at real size under the oracle (padded and folded streams)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_ak
mkdir -p $OUT
cd $R
timeout 2000 python3 tools/perforated_parity.py 120 0.4 > $OUT/perforated_n120_k0.4_with_U_parity.jsonl 2> $OUT/err.txt
echo "rc=$?"; cut -c1-1200 $OUT/perforated_n120_k0.4_with_U_parity.jsonl; tail -3 $OUT/err.txt | cut -c1-300
