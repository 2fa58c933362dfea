#!/bin/bash
# round 6, end-of-round state: smoke, the whole GPU suite, the default bench line as the driver types it (headline + secondary
# legs incl. 400^3 + CPU sample), the driver's --steps 20 --warmup 5 command, the kernel trace and the PMC passes of the headline
# workload, the N = 2 / N = 8 dry runs on one GPU over the stream-ordered stand-in.
# usage: gpurun --timeout 5000 -- 'bash tools/sessions/end_of_round_r06.sh [tag]'
R=$GRAFT_REPO_ROOT
T=${1:-final}
OUT=$R/gpurun_out/r06_$T
mkdir -p $OUT
cd $R
python3 __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.txt
timeout 3000 python3 -m pytest tests -m gpu -q --durations=20 > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.txt | cut -c1-300
timeout 1500 python3 bench.py > $OUT/bench_default_flags.json 2> $OUT/bench_err.txt
cut -c1-300 $OUT/bench_default_flags.json
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary > $OUT/bench_driver_command_steps20_warmup5.json 2>> $OUT/bench_err.txt
cut -c1-200 $OUT/bench_driver_command_steps20_warmup5.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-secondary > $OUT/bench_n148_under_rocprofv3.json 2> $OUT/bench_rocprof.err
cd $R
F=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/trace_summary.py $F > $OUT/bench_n148_kernel_trace_summary.txt 2>&1
S=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp $S $OUT/bench_n148_kernel_stats.csv
head -8 $OUT/bench_n148_kernel_trace_summary.txt
rm -rf $OUT/trace
bash tools/pmc_run.sh gpurun_out/r06_$T/pmc > $OUT/pmc_fetch_write_n148.txt 2>&1
grep -E "k_numeric|k_spmv<|k_spmv2|k_update|k_step|k_value_stream" $OUT/pmc_fetch_write_n148.txt | head
python3 tools/pmc_spmv_json.py $OUT/pmc_fetch_write_n148.txt $OUT/pmc_spmv.json 148 6698765463 "round 6, tools/sessions/end_of_round_r06.sh: two rocprofv3 --pmc passes over one step of bench.py's default workload" >> $OUT/bench_err.txt 2>&1
rm -rf $OUT/pmc/FETCH_SIZE $OUT/pmc/WRITE_SIZE
export STAN_RCCL_LIB=$R/tests/fake_rccl/libfake_rccl.so STAN_BENCH_BACKEND=gloo STAN_BENCH_DEVICE=0 GPU_MAX_HW_QUEUES=20 FAKE_RCCL_ASYNC=1
timeout 900 python3 bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu > $OUT/bench_gpus2_dry_run_on_one_gpu_async_stand_in.json 2> $OUT/bench_gpus2.err
timeout 1200 python3 bench.py --gpus 8 --steps 1 --warmup 1 --no-cpu --size 100 > $OUT/bench_gpus8_n100_dry_run_on_one_gpu_async_stand_in.json 2> $OUT/bench_gpus8.err
echo done > $OUT/done.txt
