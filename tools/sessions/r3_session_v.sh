#!/bin/bash
# round 3, session v: SpMV variant 19 (packed column offsets loaded one trip ahead) against the default (9)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_v
mkdir -p $OUT
cd $R
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "DOF/s %.3e" % d["value"], "spmv ms %.4f" % d["roofline"]["avg_launch_ms"], "frac %.3f" % d["roofline"]["frac"], "two-product ms %.4f" % d["roofline"]["two_product_avg_ms"], "its", d["config"]["cg_iterations"], "res %.6e" % d["config"]["rel_residual"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for rep in 1 2; do
for V in 9 19; do
  timeout 600 python3 bench.py --spmv-variant $V --steps 2 --warmup 1 --no-cpu > $OUT/bench_n148_variant${V}_$rep.json 2>> $OUT/err.txt
  line $OUT/bench_n148_variant${V}_$rep.json "148^3 variant $V rep $rep"
done
done
for K in 0.4 0.15; do
for V in 9 19 9 19; do
  timeout 600 python3 bench.py --size 120 --knockout $K --spmv-variant $V --steps 2 --warmup 1 --no-cpu > $OUT/bench_perf_k${K}_variant${V}.json 2>> $OUT/err.txt
  line $OUT/bench_perf_k${K}_variant${V}.json "120^3 knockout $K variant $V"
done
done
