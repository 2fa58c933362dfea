#!/bin/bash
# round 3, session e: the failing-rank peer-to-peer test (diagnosis), masked ragged slice ends in the SpMV (sigma sweep on
# the irregular meshes and the cube), IPC peer-to-peer between processes, the whole suite
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_e
mkdir -p $OUT
cd $R
for i in 1 2 3; do
  timeout 400 python3 -m pytest tests/test_gpu_round3.py -m gpu -x -q -k "failing_rank" > $OUT/pytest_failing_rank_$i.txt 2>&1
  echo "failing-rank run $i rc=$?"; tail -25 $OUT/pytest_failing_rank_$i.txt | cut -c1-300
done
timeout 900 python3 -m pytest tests/test_gpu_sharded.py -m gpu -x -q -k "peer_to_peer or bench" > $OUT/pytest_ipc.txt 2>&1
echo "ipc rc=$?"; tail -30 $OUT/pytest_ipc.txt | cut -c1-300
for sg in 1 4 32; do
  timeout 900 python3 bench.py --steps 3 --warmup 1 --no-cpu --size 120 --knockout 0.4 --sell-sigma $sg > $OUT/bench_perforated_n120_k40_sigma${sg}.json 2>> $OUT/bench_err.txt
done
for sg in 1 32; do
  timeout 900 python3 bench.py --steps 3 --warmup 1 --no-cpu --size 120 --knockout 0.15 --sell-sigma $sg > $OUT/bench_perforated_n120_k15_sigma${sg}.json 2>> $OUT/bench_err.txt
done
for sg in 1 32 1 32; do
  timeout 600 python3 bench.py --steps 4 --warmup 2 --no-cpu --sell-sigma $sg > $OUT/bench_n148_sigma${sg}_$RANDOM.json 2>> $OUT/bench_err.txt
done
cat $OUT/bench_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']; r = d['roofline']
    print(c['workload'][:58], 'sigma', c['sell_sigma'], 'pad %.4f' % c['ell_padding'], 'DOF/s %.3e' % (d['value'] or 0), 'spmv ms %.4f' % r['avg_launch_ms'], 'frac %.3f' % r['frac'], 'its', c['cg_iterations'], 'asm ms %.2f' % c['assemble_ms'], c['matrix_format'][-58:-40])
"
timeout 3000 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.txt | cut -c1-300
