#!/bin/bash
# Round 5, session l: bench.py with the PMC leg (counter traffic measured by the run itself).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05l; mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "rc $?" >> $O/bench_default.err
echo done > $O/done.txt
