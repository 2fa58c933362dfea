#!/bin/bash
# round 3, session ah: bench.py lets the placement search hold 3/4 of the free memory (default flags), lab: forced misses
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_ah
mkdir -p $OUT
cd $R
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "DOF/s %.3e" % d["value"], "spmv ms %.4f" % d["roofline"]["avg_launch_ms"], "frac %.3f" % d["roofline"]["frac"], "budget %.1f GB" % (d["placement_max_bytes"] / 1e9), d["config"]["placement_search"])
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1]).read()[-800:])
PY
}
timeout 900 python3 bench.py > $OUT/bench_default_flags.json 2>> $OUT/err.txt
line $OUT/bench_default_flags.json "default flags"
timeout 900 python3 bench.py --placement-fraction 0 --no-cpu > $OUT/bench_fraction0.json 2>> $OUT/err.txt
line $OUT/bench_fraction0.json "library default budget"
# lab build: the first 20 candidates count as not clear -> the search must go on to the 21st within the larger budget
export STAN_HIP_LIB=$R/stan_amd/csrc/build_lab/libstan_hip_lab.so STAN_LAB_PLACEMENT_FORCE_MISSES=20
timeout 900 python3 bench.py --no-cpu > $OUT/bench_forced_misses.json 2>> $OUT/err.txt
line $OUT/bench_forced_misses.json "lab: 20 forced misses, 3/4 budget"
timeout 900 python3 bench.py --no-cpu --placement-fraction 0 > $OUT/bench_forced_misses_fraction0.json 2>> $OUT/err.txt
line $OUT/bench_forced_misses_fraction0.json "lab: 20 forced misses, library budget"
tail -3 $OUT/err.txt | cut -c1-200
