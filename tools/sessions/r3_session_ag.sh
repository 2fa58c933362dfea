#!/bin/bash
# round 3, session ag: two-product kernels (k_spmv2, k_spmv_fold<.,.,2>) with a 128-VGPR budget: all four gathers of a trip together
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_ag
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -m gpu -q -k "folded or refresh or fused or two_product or sell" > $OUT/pytest_sel.txt 2>&1
echo "selected tests rc=$?"; tail -3 $OUT/pytest_sel.txt | cut -c1-300
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "DOF/s %.3e" % d["value"], "ms/step %.1f" % d["ms_per_step"], "spmv ms %.4f" % d["roofline"]["avg_launch_ms"], "frac %.3f" % d["roofline"]["frac"], "two-product ms %s" % d["roofline"]["two_product_avg_ms"], "its", d["config"]["cg_iterations"], "streams", d["config"]["repacked_streams"], "res %.6e" % d["config"]["rel_residual"])
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1]).read()[-800:])
PY
}
for rep in 1 2 3; do
  timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/bench_n148_$rep.json 2>> $OUT/err.txt
  line $OUT/bench_n148_$rep.json "148^3 default"
done
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu --fixed48 > $OUT/bench_n148_fixed48.json 2>> $OUT/err.txt
line $OUT/bench_n148_fixed48.json "148^3 fixed48"
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu --mixed > $OUT/bench_n148_mixed.json 2>> $OUT/err.txt
line $OUT/bench_n148_mixed.json "148^3 mixed"
for F in 0 -1 0 -1; do
  timeout 600 python3 bench.py --size 120 --knockout 0.4 --fold $F --steps 2 --warmup 1 --no-cpu > $OUT/bench_perf_k0.4_fold${F}_$RANDOM.json 2>> $OUT/err.txt
  line $(ls -t $OUT/bench_perf_k0.4_fold${F}_*.json | head -1) "120^3 knockout 0.4 fold $F"
done
for F in 0 -1; do
  timeout 600 python3 bench.py --size 120 --knockout 0.4 --fold $F --fixed48 --steps 2 --warmup 1 --no-cpu > $OUT/bench_perf_k0.4_fixed48_fold$F.json 2>> $OUT/err.txt
  line $OUT/bench_perf_k0.4_fixed48_fold$F.json "120^3 knockout 0.4 fixed48 fold $F"
  timeout 600 python3 bench.py --size 120 --knockout 0.25 --fold $F --steps 2 --warmup 1 --no-cpu > $OUT/bench_perf_k0.25_fold$F.json 2>> $OUT/err.txt
  line $OUT/bench_perf_k0.25_fold$F.json "120^3 knockout 0.25 fold $F"
done
