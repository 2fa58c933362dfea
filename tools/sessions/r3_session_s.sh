#!/bin/bash
# round 3, session s: ragged (re-packed) streams -- bits, then the irregular-mesh SpMV with and without them
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_s
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_round3.py -m gpu -q -x -k "ragged or sell_c_sigma" > $OUT/pytest_ragged.txt 2>&1
echo "ragged tests rc=$?"; tail -15 $OUT/pytest_ragged.txt | cut -c1-400
for K in 0.4 0.15; do
  for RG in 0 1; do
    timeout 600 python3 bench.py --size 120 --knockout $K --ragged $RG --steps 2 --warmup 1 --no-cpu > $OUT/bench_perforated_n120_k${K}_ragged$RG.json 2> $OUT/bench_k${K}_rag$RG.err
    python3 - $OUT/bench_perforated_n120_k${K}_ragged$RG.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "DOF/s %.3e" % d["value"], "spmv ms %.4f" % d["roofline"]["avg_launch_ms"], "frac %.3f" % d["roofline"]["frac"], "its", d["config"]["cg_iterations"], "ragged", d["config"]["ragged_stream"], "pad %.3f" % d["config"]["ell_padding"], "res %.3e" % d["config"]["rel_residual"])
PY
  done
done
for RG in 0 1 0 1; do
  timeout 600 python3 bench.py --ragged $RG --steps 2 --warmup 1 --no-cpu > $OUT/bench_n148_ragged${RG}_$RANDOM.json 2>> $OUT/err.txt
done
for f in $OUT/bench_n148_ragged*.json; do python3 - $f <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "DOF/s %.3e" % d["value"], "spmv ms %.4f" % d["roofline"]["avg_launch_ms"], "frac %.3f" % d["roofline"]["frac"], "its", d["config"]["cg_iterations"], "ragged", d["config"]["ragged_stream"])
PY
done
for P in "--fixed48" "--mixed"; do
  for RG in 0 1; do
    timeout 600 python3 bench.py --size 120 --knockout 0.4 --ragged $RG $P --steps 2 --warmup 1 --no-cpu > $OUT/bench_perforated_n120_k0.4${P}_ragged$RG.json 2>> $OUT/err.txt
    python3 - $OUT/bench_perforated_n120_k0.4${P}_ragged$RG.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "DOF/s %.3e" % d["value"], "spmv ms %.4f" % d["roofline"]["avg_launch_ms"], "frac %.3f" % d["roofline"]["frac"], "its", d["config"]["cg_iterations"], "ragged", d["config"]["ragged_stream"])
PY
  done
done
timeout 3000 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.txt | cut -c1-300
