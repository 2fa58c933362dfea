#!/bin/bash
# round 3, session c: full GPU suite (no -x), kernel trace + stats of the default bench, PMC passes
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c
mkdir -p $OUT
cd $R
timeout 3000 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.txt
tail -40 $OUT/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/bench_n148_under_rocprofv3.json 2> $OUT/bench_rocprof.err
cd $R
F=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/trace_summary.py $F > $OUT/bench_n148_kernel_trace_summary.txt 2>&1
S=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp $S $OUT/bench_n148_kernel_stats.csv
cat $OUT/bench_n148_kernel_trace_summary.txt
rm -rf $OUT/trace
bash tools/pmc_run.sh gpurun_out/r03_c/pmc > $OUT/pmc_fetch_write_n148.txt 2>&1
cat $OUT/pmc_fetch_write_n148.txt | tail -30
rm -rf $OUT/pmc/FETCH_SIZE $OUT/pmc/WRITE_SIZE
timeout 600 python3 bench.py --steps 5 --warmup 2 > $OUT/bench_n148_fp64_default.json 2>> $OUT/bench_err.txt
cat $OUT/bench_n148_fp64_default.json
