#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_v
mkdir -p $OUT
cd $R
timeout 2400 python3 tools/fuzz_parity.py 400 800 > $OUT/fuzz_sweep_400_1200.txt 2>&1
tail -2 $OUT/fuzz_sweep_400_1200.txt
for spec in "6 fuzz:124" "7 fuzz:101" "5 12" "8 fuzz:150"; do
  set -- $spec
  bash tools/shard_loop.sh $1 $2 2 >> $OUT/shard_loops.txt 2>&1
done
cat $OUT/shard_loops.txt
