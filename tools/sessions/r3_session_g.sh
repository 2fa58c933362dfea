#!/bin/bash
# round 3, session g: IPC peer-to-peer after the ordering fix (repeated), full suite, kernel trace of the bench,
# stan_solver phases again
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_g
mkdir -p $OUT
cd $R
for i in 1 2 3; do
  timeout 900 python3 -m pytest tests/test_gpu_sharded.py -m gpu -q -k "peer_to_peer" > $OUT/pytest_ipc_$i.txt 2>&1
  echo "ipc run $i rc=$?"; tail -4 $OUT/pytest_ipc_$i.txt | cut -c1-200
done
for i in 1 2; do
  timeout 900 python3 -m pytest tests/test_gpu_round3.py -m gpu -q -k "peer_to_peer or failing" > $OUT/pytest_p2p_$i.txt 2>&1
  echo "p2p in-process run $i rc=$?"; tail -4 $OUT/pytest_p2p_$i.txt | cut -c1-200
done
timeout 3000 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.txt | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/bench_n148_under_rocprofv3.json 2> $OUT/bench_rocprof.err
cd $R
F=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/trace_summary.py $F > $OUT/bench_n148_kernel_trace_summary.txt 2>&1
S=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
cp $S $OUT/bench_n148_kernel_stats.csv
head -28 $OUT/bench_n148_kernel_trace_summary.txt
rm -rf $OUT/trace
timeout 600 python3 bench.py --steps 5 --warmup 2 > $OUT/bench_n148_fp64_default.json 2>> $OUT/bench_err.txt
cut -c1-600 $OUT/bench_n148_fp64_default.json
timeout 1500 python3 tools/cli_scale.py 148 > $OUT/cli_scale_n148.txt 2>&1
grep -E "^\{|wall" $OUT/cli_scale_n148.txt | cut -c1-1200
