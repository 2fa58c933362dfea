#!/bin/bash
# round 3, session d: irregular-mesh SpMV (sigma 32 vs 1), stan_solver host phases at scale, peer-to-peer exchange
# times, assembly after the k_fill_cols transposition, available counters (placement question)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_d
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_gpu_subset.txt 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu_subset.txt
for sg in 32 1 32 1; do
  timeout 900 python3 bench.py --steps 3 --warmup 1 --no-cpu --size 120 --knockout 0.4 --sell-sigma $sg > $OUT/bench_perforated_n120_k40_sigma${sg}_$RANDOM.json 2>> $OUT/bench_err.txt
done
for sg in 32 1; do
  timeout 900 python3 bench.py --steps 3 --warmup 1 --no-cpu --size 120 --knockout 0.15 --sell-sigma $sg > $OUT/bench_perforated_n120_k15_sigma${sg}.json 2>> $OUT/bench_err.txt
done
cat $OUT/bench_perforated*.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']; r = d['roofline']
    print(c['workload'][:70], 'sigma', c['sell_sigma'], 'pad %.4f' % c['ell_padding'], 'DOF/s %.3e' % (d['value'] or 0), 'spmv ms %.4f' % r['avg_launch_ms'], 'frac %.3f' % r['frac'], 'its', c['cg_iterations'], 'asm ms %.1f' % c['assemble_ms'], c['matrix_format'][-58:])
"
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/bench_n148_after_fill_cols.json 2>> $OUT/bench_err.txt
python3 -c "
import json; d = json.load(open('$OUT/bench_n148_after_fill_cols.json')); print('148^3: DOF/s', d['value'], 'asm ms', d['config']['assemble_ms'], 'frac', d['roofline']['frac'])"
STAN_RCCL_LIB=$R/tests/fake_rccl/libfake_rccl.so GPU_MAX_HW_QUEUES=12 timeout 600 python3 tools/p2p_latency.py 48 2 > $OUT/p2p_latency_n48_2ranks.jsonl 2>> $OUT/p2p_err.txt
STAN_RCCL_LIB=$R/tests/fake_rccl/libfake_rccl.so GPU_MAX_HW_QUEUES=12 timeout 600 python3 tools/p2p_latency.py 48 4 > $OUT/p2p_latency_n48_4ranks.jsonl 2>> $OUT/p2p_err.txt
cat $OUT/p2p_latency_n48_2ranks.jsonl $OUT/p2p_latency_n48_4ranks.jsonl | cut -c1-400
timeout 1500 python3 tools/cli_scale.py 148 > $OUT/cli_scale_n148.txt 2>&1
tail -12 $OUT/cli_scale_n148.txt
rocprofv3 --list-avail > $OUT/rocprofv3_list_avail.txt 2>&1
grep -c . $OUT/rocprofv3_list_avail.txt; grep -i -E "TCC_EA0_RDREQ|TCC_EA0_WRREQ|DRAM|MALL|TCC_TAG_STALL|TCC_EA0_RD_UNCACHED|HBM" $OUT/rocprofv3_list_avail.txt | cut -c1-160 | head -40
