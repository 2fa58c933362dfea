#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05k; mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python3 tools/fuzz_parity.py 0 300 0 round5 > $O/fuzz_round5.txt 2>&1; echo "rc $?" >> $O/fuzz_round5.txt
echo done > $O/done.txt
