#!/bin/bash
# round 3, session p: the gathered block of x as a 16-B + 8-B load: cube and irregular meshes, A/B by two builds in one session
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_p
mkdir -p $OUT
cd $R
cp stan_amd/lib/libstan_hip.so /tmp/lib_vec.so
make -s -C stan_amd/csrc clean > /dev/null 2>&1
make -s -C stan_amd/csrc -j 16 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-value -ffp-contract=fast -DSTAN_X_VEC=0" > $OUT/build_novec.txt 2>&1
cp stan_amd/lib/libstan_hip.so /tmp/lib_novec.so
for rep in 1 2; do
for v in vec novec; do
  export STAN_HIP_LIB=/tmp/lib_$v.so
  timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/bench_n148_${v}_$rep.json 2>> $OUT/bench_err.txt
  timeout 600 python3 bench.py --steps 2 --warmup 1 --no-cpu --size 120 --knockout 0.4 > $OUT/bench_perf40_${v}_$rep.json 2>> $OUT/bench_err.txt
  timeout 600 python3 bench.py --steps 2 --warmup 1 --no-cpu --size 120 --knockout 0.15 > $OUT/bench_perf15_${v}_$rep.json 2>> $OUT/bench_err.txt
done
done
export STAN_HIP_LIB=/tmp/lib_vec.so
timeout 600 python3 bench.py --steps 2 --warmup 1 --no-cpu --fixed48 > $OUT/bench_n148_fixed48_vec.json 2>> $OUT/bench_err.txt
export STAN_HIP_LIB=/tmp/lib_novec.so
timeout 600 python3 bench.py --steps 2 --warmup 1 --no-cpu --fixed48 > $OUT/bench_n148_fixed48_novec.json 2>> $OUT/bench_err.txt
ls $OUT/bench_*.json | while read f; do python3 -c "
import json
d = json.load(open('$f')); c = d['config']; r = d['roofline']
print('$f'.split('/')[-1][:34].ljust(34), 'DOF/s %.3e' % (d['value'] or 0), 'spmv ms %.4f' % r['avg_launch_ms'], 'frac %.3f' % r['frac'], 'its', c['cg_iterations'], 'res %.6e' % c['rel_residual'])
"; done
