#!/bin/bash
# HBM traffic of the hot kernels from PMC counters (separate passes, as the guide prescribes).
# usage (on the GPU box, from the repo root): bash tools/pmc_run.sh <outdir> [bench args]
OUT=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$OUT/$C -o pmc -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-secondary "$@" > $R/$OUT/bench_$C.json 2> $R/$OUT/bench_$C.err
done
cd $R
python3 tools/pmc_summary.py $OUT
