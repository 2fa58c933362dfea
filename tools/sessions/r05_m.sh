#!/bin/bash
# Round 5, session m: the bench.py tests once more after probe_report / the PMC leg.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05m; mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python3 -m pytest tests/test_gpu_sharded.py -q -m gpu -k "bench" > $O/pytest_bench.txt 2>&1
echo "rc $?" >> $O/pytest_bench.txt
echo done > $O/done.txt
