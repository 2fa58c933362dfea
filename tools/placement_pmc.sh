#!/bin/bash
# Counter passes for the placement question (one counter per pass; program directly behind `--`).
# usage (GPU box, repo root): bash tools/placement_pmc.sh <outdir>
OUT=$1
R=$GRAFT_REPO_ROOT
export STAN_HIP_LIB=$R/stan_amd/csrc/build_lab/libstan_hip_lab.so
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/placement_pmc.py 148 10 > $R/$OUT/plain_run.txt 2>&1
for C in TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_TAG_STALL TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ_DRAM TCC_BUBBLE TCC_HIT TCC_MISS; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv json -d $R/$OUT/$C -o pmc -- python3 $R/tools/placement_pmc.py 148 10 > $R/$OUT/run_$C.txt 2> $R/$OUT/run_$C.err
  tail -1 $R/$OUT/run_$C.txt
done
cd $R
