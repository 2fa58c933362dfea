#!/bin/bash
# round 3, session ac (lab build): what do the second accumulators of the fold kernel cost?  148^3 cube, folding forced on
# (nothing is folded there): shipped kernel against a variant without the selects, against the padded kernel
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_ac
mkdir -p $OUT
cd $R
export STAN_HIP_LIB=$R/stan_amd/csrc/build_lab/libstan_hip_lab.so
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "DOF/s %.3e" % d["value"], "spmv ms %.4f" % d["roofline"]["avg_launch_ms"], "two-product ms %s" % d["roofline"]["two_product_avg_ms"], "its", d["config"]["cg_iterations"], "streams", d["config"]["repacked_streams"], "res %.6e" % d["config"]["rel_residual"])
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1]).read()[-600:])
PY
}
for rep in 1 2; do
for CFG in "0 -1" "1 -1" "1 21"; do
  set -- $CFG
  timeout 600 python3 bench.py --fold $1 --spmv-variant $2 --steps 2 --warmup 1 --no-cpu > $OUT/b.json 2>> $OUT/err.txt
  line $OUT/b.json "148^3 fold $1 variant $2"
done
done
tail -3 $OUT/err.txt | cut -c1-200
