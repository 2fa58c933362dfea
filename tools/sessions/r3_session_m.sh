#!/bin/bash
# round 3, session m: shared-gradient block arithmetic in the numeric assembly: parity suite, fuzz sweep, kernel trace
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_m
mkdir -p $OUT
cd $R
timeout 3000 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.txt | cut -c1-300
timeout 900 python3 tools/fuzz_parity.py 0 300 > $OUT/fuzz_sweep_0_300.txt 2>&1
echo "fuzz rc=$?"; tail -4 $OUT/fuzz_sweep_0_300.txt | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/bench_n148_under_rocprofv3.json 2> $OUT/bench_rocprof.err
cd $R
F=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/trace_summary.py $F > $OUT/bench_n148_kernel_trace_summary.txt 2>&1
S=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp $S $OUT/bench_n148_kernel_stats.csv
head -14 $OUT/bench_n148_kernel_trace_summary.txt
rm -rf $OUT/trace
bash tools/pmc_run.sh gpurun_out/r03_m/pmc > $OUT/pmc_fetch_write_n148.txt 2>&1
grep -E "k_numeric|k_spmv<|k_symbolic|k_fill_cols" $OUT/pmc_fetch_write_n148.txt | head
rm -rf $OUT/pmc/FETCH_SIZE $OUT/pmc/WRITE_SIZE
timeout 600 python3 bench.py --steps 5 --warmup 2 > $OUT/bench_n148_fp64_default.json 2>> $OUT/bench_err.txt
cut -c1-700 $OUT/bench_n148_fp64_default.json
