#!/bin/bash
# Round 5, session d: the whole GPU suite (all failures, no -x) on the restructured CG + the new export path of the
# console driver (results kept on the device, mapped by the writer threads, parallel pwrite, mmap reader, prewarmed
# context); the driver's phases at 148^3, twice.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 600 python3 tools/cli_scale.py 148 > $O/cli_scale_n148_a.txt 2>&1
timeout 600 python3 tools/cli_scale.py 148 > $O/cli_scale_n148_b.txt 2>&1
timeout 3600 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1
echo "rc $?" >> $O/pytest_gpu.txt
echo done > $O/done.txt
