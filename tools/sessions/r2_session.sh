#!/bin/bash
# Round-2 GPU session: placement map (lab build), parity prints, rocprofv3 kernel trace + PMC passes.
# usage (GPU box, repo root): bash tools/sessions/r2_session.sh <tag>
TAG=${1:-a}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_$TAG
mkdir -p $OUT
cd $R
rocprofv3 -L > $OUT/counters_list.txt 2>&1
STAN_HIP_LIB=$R/stan_amd/csrc/build_lab/libstan_hip_lab.so timeout 900 python3 tools/placement_map.py 148 8 16 1 > $OUT/placement_map_n148.txt 2>&1
timeout 900 python3 -m pytest tests/test_gpu_round2.py -m gpu -q -s -k "bench_mode or config3 or config5" > $OUT/parity_prints.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu > $OUT/bench_under_rocprofv3.json 2> $OUT/bench_under_rocprofv3.err
cd $R
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_summary.py $F > $OUT/kernel_trace_summary.txt 2>&1
S=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp $S $OUT/kernel_stats.csv
rm -rf $OUT/trace
bash tools/pmc_run.sh gpurun_out/r02_$TAG/pmc > $OUT/pmc_summary_stdout.txt 2>&1
find $OUT/pmc -name "*.csv" -size +2M -delete
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --single-reduce --no-cpu > $OUT/bench_single_reduce.json 2> $OUT/bench_single_reduce.err
