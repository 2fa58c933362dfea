#!/bin/bash
# Round 5, session f: the console driver with the mapped export (3 runs + 1 with write calls), config 5's arithmetic at
# 400^3 said honestly (fp32 copy refined to 1e-8 in fp64 terms, both refine modes; cg_ms is the figure: one step, no
# warm-up), kernel trace of the default bench, the new tests.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f; mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for i in a b c; do timeout 600 python3 tools/cli_scale.py 148 > $O/cli_scale_n148_$i.txt 2>&1; done
STAN_STDB_WRITE=pwrite timeout 600 python3 tools/cli_scale.py 148 > $O/cli_scale_n148_pwrite.txt 2>&1
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_formats.py tests/test_gpu_console.py -q -m gpu -x > $O/pytest_subset.txt 2>&1
echo "rc $?" >> $O/pytest_subset.txt
timeout 1500 python3 bench.py --size 400 --mixed --refine 2 --steps 1 --warmup 0 --no-cpu > $O/bench_n400_mixed_refine2.json 2> $O/bench_n400_mixed_refine2.err
timeout 1500 python3 bench.py --size 400 --mixed --refine 1 --steps 1 --warmup 0 --no-cpu > $O/bench_n400_mixed_refine1.json 2> $O/bench_n400_mixed_refine1.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace -o run -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-secondary > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cd $R
python3 tools/trace_summary.py $O/bench_trace/run_kernel_trace.csv > $O/bench_trace_summary.txt 2>&1
rm -f $O/bench_trace/run_kernel_trace.csv
echo done > $O/done.txt
