#!/bin/bash
# round 3, session b: the whole GPU suite on the SELL-C-sigma / peer-to-peer tree, then the bench line with
# the sorting window on (32) and off (1)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_b
mkdir -p $OUT
cd $R
timeout 2400 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.txt
tail -30 $OUT/pytest_gpu.txt
for sg in 32 1 32 1; do
  timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cpu --sell-sigma $sg > $OUT/bench_n148_sigma${sg}_$RANDOM.json 2> $OUT/bench_err.txt
done
cat $OUT/bench_n148_sigma*.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config']['sell_sigma'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['config']['ell_padding'], d['config']['assemble_ms'], d['config']['matrix_format'][-60:])
"
