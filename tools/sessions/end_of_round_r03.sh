#!/bin/bash
# round 3, session final: end-of-round state (after the irregular-mesh work): smoke, full suite, bench line, kernel trace, PMC passes, 200^3 line
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_final
mkdir -p $OUT
cd $R
python3 __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.txt
timeout 3000 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu.txt | cut -c1-300
timeout 900 python3 bench.py > $OUT/bench_default_flags.json 2> $OUT/bench_err.txt
cut -c1-400 $OUT/bench_default_flags.json
timeout 900 python3 bench.py --steps 5 --warmup 2 > $OUT/bench_n148_fp64_final.json 2>> $OUT/bench_err.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/bench_n148_under_rocprofv3_final.json 2> $OUT/bench_rocprof.err
cd $R
F=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/trace_summary.py $F > $OUT/bench_n148_kernel_trace_summary_final.txt 2>&1
S=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp $S $OUT/bench_n148_kernel_stats_final.csv
head -8 $OUT/bench_n148_kernel_trace_summary_final.txt
rm -rf $OUT/trace
bash tools/pmc_run.sh gpurun_out/r03_final/pmc > $OUT/pmc_fetch_write_n148_final.txt 2>&1
grep -E "k_numeric|k_spmv<|k_spmv2|k_update|k_step" $OUT/pmc_fetch_write_n148_final.txt | head
rm -rf $OUT/pmc/FETCH_SIZE $OUT/pmc/WRITE_SIZE
timeout 900 python3 bench.py --steps 3 --warmup 1 --no-cpu --size 200 > $OUT/bench_n200_fp64_final.json 2>> $OUT/bench_err.txt
timeout 900 python3 bench.py --steps 3 --warmup 1 --no-cpu --fixed48 > $OUT/bench_n148_fixed48_final.json 2>> $OUT/bench_err.txt
timeout 900 python3 bench.py --steps 3 --warmup 1 --no-cpu --mixed > $OUT/bench_n148_mixed_final.json 2>> $OUT/bench_err.txt
for f in bench_n148_fp64_final bench_n200_fp64_final bench_n148_fixed48_final bench_n148_mixed_final; do python3 -c "
import json; d = json.load(open('$OUT/$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['config']['assemble_ms'], d['config']['cg_iterations'])"; done
timeout 900 python3 bench.py --size 120 --knockout 0.4 --steps 3 --warmup 1 --no-cpu > $OUT/bench_perforated_n120_k0.4_final.json 2>> $OUT/bench_err.txt
timeout 900 python3 bench.py --size 120 --knockout 0.4 --fold 0 --steps 3 --warmup 1 --no-cpu > $OUT/bench_perforated_n120_k0.4_fold0_final.json 2>> $OUT/bench_err.txt
for f in bench_perforated_n120_k0.4_final bench_perforated_n120_k0.4_fold0_final; do python3 -c "
import json; d = json.load(open('$OUT/$f.json')); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['config']['repacked_streams'], d['config']['cg_iterations'])"; done
