#!/bin/bash
# round 3, session al: the console driver end to end on an irregular mesh at scale (120^3 box, 40 % of the elements removed)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_al
mkdir -p $OUT
cd $R
timeout 1200 python3 tools/cli_scale.py 120 0.4 > $OUT/cli_scale_perforated_n120_k0.4.txt 2>&1
echo "rc=$?"; tail -12 $OUT/cli_scale_perforated_n120_k0.4.txt | cut -c1-1200
