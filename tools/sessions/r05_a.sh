#!/bin/bash
# Round 5, first GPU session: config 4 at size (8 ranks of one process on one GPU, both transports), k_recover timed and
# counted, 400^3 mixed with the pool budget, the console driver's phases at 148^3.
# usage (GPU box, repo root): bash tools/sessions/r05_a.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05a; mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python3 -m pytest tests/test_gpu_multi.py -k config4 -x -q -s > $O/config4_test.txt 2>&1
echo "config4 rc $?" >> $O/config4_test.txt
# k_recover: wall, kernel trace, FETCH / WRITE passes
timeout 300 python3 tools/recover_time.py 148 10 > $O/recover_time_n148.json 2> $O/recover_time.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/recover_trace -o run -- python3 $R/tools/recover_time.py 148 10 > $O/recover_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/recover_pmc/$C -o pmc -- python3 $R/tools/recover_time.py 148 3 > $O/recover_pmc_$C.log 2>&1
done
cd $R
python3 tools/pmc_summary.py gpurun_out/r05a/recover_pmc > $O/recover_pmc_summary.txt 2>&1
python3 tools/trace_summary.py $O/recover_trace/run_kernel_trace.csv > $O/recover_trace_summary.txt 2>&1
# the console driver at 148^3 (phases in its --json line)
timeout 600 python3 tools/cli_scale.py 148 > $O/cli_scale_n148.txt 2>&1
# 400^3 mixed with the pool budget of bench.py (0.9 of free)
timeout 900 python3 bench.py --size 400 --mixed --steps 2 --warmup 1 --no-cpu > $O/bench_n400_mixed_pool90.json 2> $O/bench_n400_mixed.err
echo done > $O/done.txt
