#!/bin/bash
# Round 5, session e: the console driver with the fast result encoder and the parallel incidence build; bench.py as the
# driver types it (headline + secondary legs); the N > 1 line with its probe children as a dry run on one GPU; the GPU
# suite (all failures).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05e; mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python3 tools/cli_scale.py 148 > $O/cli_scale_n148_a.txt 2>&1
timeout 600 python3 tools/cli_scale.py 148 > $O/cli_scale_n148_b.txt 2>&1
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "rc $?" >> $O/bench_default.err
STAN_RCCL_LIB=$R/tests/fake_rccl/libfake_rccl.so STAN_BENCH_BACKEND=gloo STAN_BENCH_DEVICE=0 timeout 900 python3 bench.py --gpus 2 --steps 1 --warmup 1 --size 100 --no-cpu > $O/bench_gpus2_dry_run.json 2> $O/bench_gpus2_dry_run.err
echo "rc $?" >> $O/bench_gpus2_dry_run.err
timeout 3600 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
echo "rc $?" >> $O/pytest_gpu.txt
echo done > $O/done.txt
