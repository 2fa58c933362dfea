#!/bin/bash
# Round 5, session o: 400^3 steady-state lines on the final tree (one warm-up step, one timed step): the fp32 copy refined to
# 1e-8 in fp64 terms (refine 2), FIXED-48, fp64 -- assemble_ms without allocator time (pool budget 0.9 x free).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05o; mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1800 python3 bench.py --size 400 --mixed --refine 2 --steps 1 --warmup 1 --no-cpu > $O/bench_n400_mixed.json 2> $O/bench_n400_mixed.err
timeout 1500 python3 bench.py --size 400 --fixed48 --steps 1 --warmup 1 --no-cpu > $O/bench_n400_fixed48.json 2> $O/bench_n400_fixed48.err
timeout 1500 python3 bench.py --size 400 --steps 1 --warmup 1 --no-cpu > $O/bench_n400_fp64.json 2> $O/bench_n400_fp64.err
echo done > $O/done.txt
