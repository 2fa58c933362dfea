#!/bin/bash
# round 3, session n: the balanced accumulation (byte map) in the numeric assembly: parity + kernel trace
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_n
mkdir -p $OUT
cd $R
timeout 3000 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.txt | cut -c1-300
timeout 900 python3 tools/fuzz_parity.py 0 300 > $OUT/fuzz_sweep_0_300.txt 2>&1
echo "fuzz rc=$?"; tail -2 $OUT/fuzz_sweep_0_300.txt | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/bench_n148_under_rocprofv3.json 2> $OUT/bench_rocprof.err
cd $R
F=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/trace_summary.py $F > $OUT/bench_n148_kernel_trace_summary.txt 2>&1
S=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp $S $OUT/bench_n148_kernel_stats.csv
head -14 $OUT/bench_n148_kernel_trace_summary.txt
rm -rf $OUT/trace
