#!/bin/bash
# round 3, session u: streaming-pattern microbenchmark (bytes per lane, partial waves, alignment) + ragged tests
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_u
mkdir -p $OUT
cd $R
timeout 300 tools/lab/stream_lab 6 > $OUT/stream_lab.txt 2>&1
cat $OUT/stream_lab.txt | cut -c1-200
timeout 300 tools/lab/stream_lab 6 > $OUT/stream_lab_b.txt 2>&1
timeout 900 python3 -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "ragged" > $OUT/pytest_ragged.txt 2>&1
echo "ragged tests rc=$?"; tail -8 $OUT/pytest_ragged.txt | cut -c1-400
