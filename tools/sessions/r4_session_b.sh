#!/bin/bash
# round 4, session b: the two fixed tests, k_numeric A/B (LDS accumulator vs register form, XCD-chunked mapping), kernel trace and
# PMC passes of the bench, the 400^3 config-5 test, the 200^3 line (two-base packed columns), the driver's N = 2 command as a
# dry run on one GPU (launcher + transport probe)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_b
mkdir -p $OUT
cd $R
timeout 600 python3 -m pytest tests/test_gpu_round4.py tests/test_gpu_sharded.py -m gpu -q -k "wedge or survivor" > $OUT/pytest_fixed.txt 2>&1
echo "fixed rc=$?"; tail -6 $OUT/pytest_fixed.txt | cut -c1-300
timeout 900 python3 tools/numeric_variants.py 148 0,1,2,3 > $OUT/numeric_variants_n148.txt 2>&1; grep variant $OUT/numeric_variants_n148.txt
cd /tmp && export TMPDIR=/tmp
for V in 0 1; do
  STAN_NUMERIC_VARIANT=$V rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$V -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/bench_n148_under_rocprofv3_variant$V.json 2> $OUT/bench_rocprof$V.err
  F=$(find $OUT/trace$V -name '*kernel_trace.csv' | head -1)
  python3 $R/tools/trace_summary.py $F > $OUT/bench_n148_kernel_trace_summary_variant$V.txt 2>&1
  head -12 $OUT/bench_n148_kernel_trace_summary_variant$V.txt
  rm -rf $OUT/trace$V
done
cd $R
bash tools/pmc_run.sh gpurun_out/r04_b/pmc0 > $OUT/pmc_fetch_write_n148_variant0.txt 2>&1
STAN_NUMERIC_VARIANT=1 bash tools/pmc_run.sh gpurun_out/r04_b/pmc1 > $OUT/pmc_fetch_write_n148_variant1.txt 2>&1
grep -E "k_numeric|k_symbolic|k_spmv<|k_spmv2" $OUT/pmc_fetch_write_n148_variant0.txt $OUT/pmc_fetch_write_n148_variant1.txt | head -12
rm -rf $OUT/pmc0/FETCH_SIZE $OUT/pmc0/WRITE_SIZE $OUT/pmc1/FETCH_SIZE $OUT/pmc1/WRITE_SIZE
timeout 900 python3 bench.py --steps 3 --warmup 1 --no-cpu --size 200 > $OUT/bench_n200_fp64.json 2>> $OUT/bench_err.txt
python3 -c "
import json; d = json.load(open('$OUT/bench_n200_fp64.json')); print('n200', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['config']['matrix_format'])"
STAN_RCCL_LIB=$R/tests/fake_rccl/libfake_rccl.so STAN_BENCH_BACKEND=gloo STAN_BENCH_DEVICE=0 timeout 900 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu > $OUT/bench_gpus2_no_launcher_dry_run_on_one_gpu.json 2> $OUT/bench_gpus2.err
echo "gpus2 rc=$?"; cut -c1-300 $OUT/bench_gpus2_no_launcher_dry_run_on_one_gpu.json
python3 -c "
import json; d = json.load(open('$OUT/bench_gpus2_no_launcher_dry_run_on_one_gpu.json')); print(json.dumps(d['config'].get('p2p_probe'))[:1500]); print(d['config'].get('recommended_transport'))"
STAN_RCCL_LIB=$R/tests/fake_rccl/libfake_rccl.so STAN_BENCH_DEVICE=0 timeout 900 python3 bench.py --gpus 2 --one-process --steps 2 --warmup 1 > $OUT/bench_one_process_2ranks_on_one_gpu.json 2> $OUT/bench_onep.err
echo "one-process rc=$?"; cut -c1-400 $OUT/bench_one_process_2ranks_on_one_gpu.json
timeout 1500 python3 -m pytest tests/test_gpu_round4.py -m gpu -q -s -k "400_cubed" > $OUT/pytest_400.txt 2>&1
echo "400 rc=$?"; grep -E "400\^3|passed|failed|Error|assert" $OUT/pytest_400.txt | head -10
