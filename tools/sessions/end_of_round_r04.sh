#!/bin/bash
# round 4, end-of-round state: smoke, the whole GPU suite, the default bench line (with the CPU sample), a 5-step line, the kernel
# trace and the PMC passes of the same command, the 200^3 / FIXED-48 / fp32-matrix lines, the irregular mesh
R=$GRAFT_REPO_ROOT
T=${1:-final}
OUT=$R/gpurun_out/r04_$T
mkdir -p $OUT
cd $R
python3 __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.txt
timeout 3000 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.txt | cut -c1-300
timeout 900 python3 bench.py > $OUT/bench_default_flags.json 2> $OUT/bench_err.txt
cut -c1-300 $OUT/bench_default_flags.json
timeout 900 python3 bench.py --steps 5 --warmup 2 --no-cpu > $OUT/bench_n148_fp64.json 2>> $OUT/bench_err.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/bench_n148_under_rocprofv3.json 2> $OUT/bench_rocprof.err
cd $R
F=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/trace_summary.py $F > $OUT/bench_n148_kernel_trace_summary.txt 2>&1
S=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp $S $OUT/bench_n148_kernel_stats.csv
head -8 $OUT/bench_n148_kernel_trace_summary.txt
rm -rf $OUT/trace
bash tools/pmc_run.sh gpurun_out/r04_$T/pmc > $OUT/pmc_fetch_write_n148.txt 2>&1
grep -E "k_numeric|k_spmv<|k_spmv2|k_update|k_step" $OUT/pmc_fetch_write_n148.txt | head
rm -rf $OUT/pmc/FETCH_SIZE $OUT/pmc/WRITE_SIZE
timeout 900 python3 bench.py --steps 3 --warmup 1 --no-cpu --size 200 > $OUT/bench_n200_fp64.json 2>> $OUT/bench_err.txt
timeout 900 python3 bench.py --steps 3 --warmup 1 --no-cpu --fixed48 > $OUT/bench_n148_fixed48.json 2>> $OUT/bench_err.txt
timeout 900 python3 bench.py --steps 3 --warmup 1 --no-cpu --mixed > $OUT/bench_n148_mixed.json 2>> $OUT/bench_err.txt
timeout 900 python3 bench.py --size 120 --knockout 0.4 --steps 3 --warmup 1 --no-cpu > $OUT/bench_perforated_n120_k0.4.json 2>> $OUT/bench_err.txt
for f in bench_n148_fp64 bench_n200_fp64 bench_n148_fixed48 bench_n148_mixed bench_perforated_n120_k0.4; do python3 -c "
import json; d = json.load(open('$OUT/$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['config']['assemble_ms'], d['config']['cg_iterations'])"; done
