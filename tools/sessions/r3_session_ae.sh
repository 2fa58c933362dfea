#!/bin/bash
# round 3, session ae: 400^3 (193 M DOF, BASELINE config 5 size, HEX8_G2 fp64) on one GPU with the end-of-round library
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_ae
mkdir -p $OUT
cd $R
t0=$(date +%s)
timeout 1700 python3 bench.py --size 400 --steps 1 --warmup 0 --no-cpu --watchdog 1600 > $OUT/bench_n400_fp64_r03.json 2> $OUT/err_fp64.txt
echo "rc=$? wall $(( $(date +%s) - t0 )) s"
python3 -c "
import json; d = json.loads(open('$OUT/bench_n400_fp64_r03.json').read().strip().splitlines()[-1]); c = d['config']
print('400^3 fp64', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'spmv ms', d['roofline']['avg_launch_ms'], 'GB', d['roofline']['bytes_per_launch'] / 1e9, 'assemble ms', c['assemble_ms'], 'its', c['cg_iterations'], 'res', c['rel_residual'], c['placement_search'])"
t0=$(date +%s)
timeout 1500 python3 bench.py --size 400 --steps 1 --warmup 0 --no-cpu --fixed48 --watchdog 1400 > $OUT/bench_n400_fixed48_r03.json 2> $OUT/err_fx48.txt
echo "rc=$? wall $(( $(date +%s) - t0 )) s"
python3 -c "
import json; d = json.loads(open('$OUT/bench_n400_fixed48_r03.json').read().strip().splitlines()[-1]); c = d['config']
print('400^3 fixed48', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'spmv ms', d['roofline']['avg_launch_ms'], 'its', c['cg_iterations'], 'res', c['rel_residual'])"
tail -2 $OUT/err_fp64.txt | cut -c1-200
