#!/bin/bash
# round 3, session l: SpMV kernel variants on the irregular mesh (lab build: occupancy caps), then the 200^3 oracle line
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_l
mkdir -p $OUT
cd $R
export STAN_HIP_LIB=$R/stan_amd/csrc/build_lab/libstan_hip_lab.so
for v in 9 0 12 13 17 18 9; do
  timeout 600 python3 bench.py --steps 2 --warmup 1 --no-cpu --size 120 --knockout 0.4 --spmv-variant $v > $OUT/bench_perf40_variant${v}_$RANDOM.json 2>> $OUT/bench_err.txt
done
for v in 9 17 18; do
  timeout 600 python3 bench.py --steps 2 --warmup 1 --no-cpu --size 120 --knockout 0.15 --spmv-variant $v > $OUT/bench_perf15_variant${v}.json 2>> $OUT/bench_err.txt
done
ls $OUT/*.json | while read f; do python3 -c "
import sys, json
d = json.load(open('$f')); c = d['config']; r = d['roofline']
print('$f'.split('/')[-1][:32], 'DOF/s %.3e' % (d['value'] or 0), 'spmv ms %.4f' % r['avg_launch_ms'], 'frac %.3f' % r['frac'], 'its', c['cg_iterations'])
"; done
unset STAN_HIP_LIB
timeout 3400 python3 tools/cpu_sizes.py 200 > $OUT/cpu_sizes_n200_with_U_parity.jsonl 2> $OUT/cpu_sizes_n200.err
echo "cpu_sizes 200 rc=$?"; cat $OUT/cpu_sizes_n200_with_U_parity.jsonl | cut -c1-1500
