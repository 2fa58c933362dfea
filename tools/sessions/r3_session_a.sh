#!/bin/bash
# round 3, session a: hipStreamWaitValue64 probe + the 148^3 oracle-vs-GPU parity line
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_a
mkdir -p $OUT
cd $R
nproc > $OUT/host.txt; free -g >> $OUT/host.txt
timeout 120 tools/lab/waitvalue_probe 8 > $OUT/waitvalue_probe_default.txt 2>&1
echo "rc=$?" >> $OUT/waitvalue_probe_default.txt
GPU_MAX_HW_QUEUES=16 timeout 120 tools/lab/waitvalue_probe 8 > $OUT/waitvalue_probe_hwq16.txt 2>&1
echo "rc=$?" >> $OUT/waitvalue_probe_hwq16.txt
cat $OUT/waitvalue_probe_default.txt $OUT/waitvalue_probe_hwq16.txt
timeout 2400 python3 tools/cpu_sizes.py 148 > $OUT/cpu_sizes_n148_with_U_parity.jsonl 2> $OUT/cpu_sizes_n148.err
echo "cpu_sizes rc=$?"
cat $OUT/cpu_sizes_n148_with_U_parity.jsonl; tail -3 $OUT/cpu_sizes_n148.err
