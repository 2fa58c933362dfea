#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05n; mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "recovery" > $O/pytest_recovery.txt 2>&1
echo "rc $?" >> $O/pytest_recovery.txt
