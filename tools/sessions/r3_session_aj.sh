#!/bin/bash
# round 3, session aj: k_step with four elements per trip -- same bits?  kernel time from the trace
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_aj
mkdir -p $OUT
cd $R
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "DOF/s %.3e" % d["value"], "ms/step %.1f" % d["ms_per_step"], "spmv ms %.4f" % d["roofline"]["avg_launch_ms"], "frac %.3f" % d["roofline"]["frac"], "its", d["config"]["cg_iterations"], "res %.6e" % d["config"]["rel_residual"])
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1]).read()[-800:])
PY
}
for rep in 1 2; do
  timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/bench_n148_$rep.json 2>> $OUT/err.txt
  line $OUT/bench_n148_$rep.json "148^3 default"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
cd $R
F=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/trace_summary.py $F > $OUT/trace_summary.txt 2>&1
head -6 $OUT/trace_summary.txt
rm -rf $OUT/trace
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q -x > $OUT/pytest_parity.txt 2>&1
echo "parity tests rc=$?"; grep -n "passed\|failed" $OUT/pytest_parity.txt | tail -2
timeout 600 python3 bench.py --size 200 --steps 2 --warmup 1 --no-cpu > $OUT/bench_n200.json 2>> $OUT/err.txt
line $OUT/bench_n200.json "200^3"
