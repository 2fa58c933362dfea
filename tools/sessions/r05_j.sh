#!/bin/bash
# Round 5, session j: fuzz sweeps on the final tree -- the paths round 5 added (first product scaling the matrix against the
# scaling pass bit for bit; the fp32 copy refined, its reported residual against an independent one) and the sweeps of
# rounds 3-4 as regression (box, collapsed, revolved, folded).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05j; mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python3 tools/fuzz_parity.py 0 300 0 round5 > $O/fuzz_round5.txt 2>&1; echo "rc $?" >> $O/fuzz_round5.txt
timeout 1200 python3 tools/fuzz_parity.py 0 300 > $O/fuzz_box.txt 2>&1; echo "rc $?" >> $O/fuzz_box.txt
timeout 900 python3 tools/fuzz_parity.py 0 150 0 collapsed > $O/fuzz_collapsed.txt 2>&1; echo "rc $?" >> $O/fuzz_collapsed.txt
timeout 900 python3 tools/fuzz_parity.py 0 60 0 revolved > $O/fuzz_revolved.txt 2>&1; echo "rc $?" >> $O/fuzz_revolved.txt
timeout 900 python3 tools/fuzz_parity.py 0 150 1 > $O/fuzz_folded.txt 2>&1; echo "rc $?" >> $O/fuzz_folded.txt
echo done > $O/done.txt
