#!/bin/bash
# Round 5, session g: the console driver (write calls again, level-parallel walk, parallel key index; stage times with
# STAN_HOST_TRACE), 400^3 FIXED-48 and fp64 next to the refined fp32 copy of session f (one step, no warm-up: cg_ms is
# the figure), the whole GPU suite.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05g; mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
for i in a b c; do STAN_HOST_TRACE=1 timeout 600 python3 tools/cli_scale.py 148 > $O/cli_scale_n148_$i.txt 2>&1; done
timeout 1500 python3 bench.py --size 400 --fixed48 --steps 1 --warmup 0 --no-cpu > $O/bench_n400_fixed48.json 2> $O/bench_n400_fixed48.err
timeout 1500 python3 bench.py --size 400 --steps 1 --warmup 0 --no-cpu > $O/bench_n400_fp64.json 2> $O/bench_n400_fp64.err
timeout 3600 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
echo "rc $?" >> $O/pytest_gpu.txt
echo done > $O/done.txt
