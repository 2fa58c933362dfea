#!/bin/bash
# round 3, session x: folded rows -- tests, then the irregular-mesh SpMV with and without them
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_x
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_round3.py -m gpu -q -k "folded or ragged" > $OUT/pytest_folded.txt 2>&1
echo "folded tests rc=$?"; tail -25 $OUT/pytest_folded.txt | cut -c1-300
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "DOF/s %.3e" % d["value"], "spmv ms %.4f" % d["roofline"]["avg_launch_ms"], "frac %.3f" % d["roofline"]["frac"], "two-product ms %s" % d["roofline"]["two_product_avg_ms"], "its", d["config"]["cg_iterations"], "streams", d["config"]["repacked_streams"], "res %.6e" % d["config"]["rel_residual"])
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1]).read()[-800:])
PY
}
for K in 0.4 0.15; do
for F in 0 1 0 1; do
  timeout 600 python3 bench.py --size 120 --knockout $K --fold $F --steps 2 --warmup 1 --no-cpu > $OUT/bench_perforated_n120_k${K}_fold${F}_$RANDOM.json 2>> $OUT/err.txt
  line $(ls -t $OUT/bench_perforated_n120_k${K}_fold${F}_*.json | head -1) "120^3 knockout $K fold $F"
done
done
for F in 0 1 -1; do
  timeout 600 python3 bench.py --fold $F --steps 2 --warmup 1 --no-cpu > $OUT/bench_n148_fold$F.json 2>> $OUT/err.txt
  line $OUT/bench_n148_fold$F.json "148^3 fold $F"
done
for P in "--fixed48" "--mixed"; do
for F in 0 1; do
  timeout 600 python3 bench.py --size 120 --knockout 0.4 --fold $F $P --steps 2 --warmup 1 --no-cpu > $OUT/bench_perforated_n120_k0.4${P}_fold$F.json 2>> $OUT/err.txt
  line $OUT/bench_perforated_n120_k0.4${P}_fold$F.json "120^3 knockout 0.4 $P fold $F"
done
done
timeout 3000 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; grep -n "passed\|failed" $OUT/pytest_gpu.txt | tail -3
