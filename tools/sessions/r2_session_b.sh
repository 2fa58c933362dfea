#!/bin/bash
# Round-2 GPU session b: new tests, walk-order variants on candidate blocks, PMC passes.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_b
mkdir -p $OUT
cd $R
timeout 1200 python3 -m pytest tests/test_gpu_multi.py tests/test_gpu_round2.py tests/test_gpu_sharded.py -m gpu -x -q > $OUT/pytest_new.txt 2>&1
tail -15 $OUT/pytest_new.txt
export STAN_HIP_LIB=$R/stan_amd/csrc/build_lab/libstan_hip_lab.so
timeout 600 python3 tools/placement_variants.py 148 8 > $OUT/placement_variants_n148.txt 2>&1
timeout 600 python3 tools/placement_variants.py 200 4 > $OUT/placement_variants_n200.txt 2>&1
unset STAN_HIP_LIB
bash tools/pmc_run.sh gpurun_out/r02_b/pmc > $OUT/pmc_summary_stdout.txt 2>&1
find $OUT/pmc -name "*.csv" -size +2M -delete
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu > $OUT/bench_under_rocprofv3.json 2> $OUT/bench_under_rocprofv3.err
cd $R
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_summary.py $F > $OUT/kernel_trace_summary.txt 2>&1
rm -rf $OUT/trace
cat $OUT/placement_variants_n148.txt
