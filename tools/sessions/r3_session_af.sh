#!/bin/bash
# round 3, session af: fold kernel with both gathers of a trip issued together (96-VGPR budget, offsets one trip ahead)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_af
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_round3.py -m gpu -q -k "folded" > $OUT/pytest_folded.txt 2>&1
echo "folded tests rc=$?"; tail -4 $OUT/pytest_folded.txt | cut -c1-300
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "DOF/s %.3e" % d["value"], "spmv ms %.4f" % d["roofline"]["avg_launch_ms"], "frac %.3f" % d["roofline"]["frac"], "two-product ms %s" % d["roofline"]["two_product_avg_ms"], "its", d["config"]["cg_iterations"], "streams", d["config"]["repacked_streams"], "res %.6e" % d["config"]["rel_residual"])
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1]).read()[-800:])
PY
}
for F in 0 1 0 1; do
  timeout 600 python3 bench.py --fold $F --steps 2 --warmup 1 --no-cpu > $OUT/b.json 2>> $OUT/err.txt
  line $OUT/b.json "148^3 fold $F"
done
for F in 0 -1 0 -1; do
  timeout 600 python3 bench.py --size 120 --knockout 0.4 --fold $F --steps 2 --warmup 1 --no-cpu > $OUT/b.json 2>> $OUT/err.txt
  line $OUT/b.json "120^3 knockout 0.4 fold $F"
done
for F in 0 -1; do
  timeout 600 python3 bench.py --size 120 --knockout 0.4 --fold $F --fixed48 --steps 2 --warmup 1 --no-cpu > $OUT/b.json 2>> $OUT/err.txt
  line $OUT/b.json "120^3 knockout 0.4 fixed48 fold $F"
  timeout 600 python3 bench.py --size 120 --knockout 0.25 --fold $F --steps 2 --warmup 1 --no-cpu > $OUT/b.json 2>> $OUT/err.txt
  line $OUT/b.json "120^3 knockout 0.25 fold $F"
done
