#!/bin/bash
# round 3, session y: folded rows as the default (auto), their value block allocated by the placement search;
# full GPU suite, smoke, default bench, irregular-mesh lines
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_y
mkdir -p $OUT
cd $R
timeout 3000 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; grep -n "passed\|failed" $OUT/pytest_gpu.txt | tail -3; grep -n "^FAILED\|Error" $OUT/pytest_gpu.txt | head -10
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ps = d["config"].get("placement_search", {})
    print(sys.argv[2], "DOF/s %.3e" % d["value"], "spmv ms %.4f" % d["roofline"]["avg_launch_ms"], "frac %.3f" % d["roofline"]["frac"], "two-product ms %s" % d["roofline"]["two_product_avg_ms"], "its", d["config"]["cg_iterations"], "streams", d["config"]["repacked_streams"], "res %.6e" % d["config"]["rel_residual"], "candidates", ps.get("candidates_timed"))
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1]).read()[-800:])
PY
}
timeout 900 python3 bench.py > $OUT/bench_default_flags.json 2>> $OUT/err.txt
line $OUT/bench_default_flags.json "default flags"
for rep in 1 2; do
for F in 0 -1; do
  timeout 600 python3 bench.py --size 120 --knockout 0.4 --fold $F --steps 2 --warmup 1 --no-cpu > $OUT/bench_perforated_n120_k0.4_fold${F}_$rep.json 2>> $OUT/err.txt
  line $OUT/bench_perforated_n120_k0.4_fold${F}_$rep.json "120^3 knockout 0.4 fold $F"
done
done
for F in 0 -1; do
  timeout 600 python3 bench.py --size 120 --knockout 0.4 --fold $F --fixed48 --steps 2 --warmup 1 --no-cpu > $OUT/bench_perforated_n120_k0.4_fixed48_fold$F.json 2>> $OUT/err.txt
  line $OUT/bench_perforated_n120_k0.4_fixed48_fold$F.json "120^3 knockout 0.4 fixed48 fold $F"
  timeout 600 python3 bench.py --size 120 --knockout 0.25 --fold $F --steps 2 --warmup 1 --no-cpu > $OUT/bench_perforated_n120_k0.25_fold$F.json 2>> $OUT/err.txt
  line $OUT/bench_perforated_n120_k0.25_fold$F.json "120^3 knockout 0.25 fold $F"
done
