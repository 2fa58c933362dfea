#!/bin/bash
# Round 5, session b: the restructured CG (cg_run: set-up / pass / finish) and the honest reduced-precision modes --
# the new precision tests first, the refine-mode comparison at 148^3, then the whole GPU suite.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1200 python3 -m pytest tests/test_gpu_precision.py -x -q -s > $O/precision_tests.txt 2>&1
echo "rc $?" >> $O/precision_tests.txt
timeout 900 python3 tools/mixed_refine.py 148 > $O/mixed_refine_n148.jsonl 2> $O/mixed_refine.err
timeout 2400 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1
echo "rc $?" >> $O/pytest_gpu.txt
echo done > $O/done.txt
