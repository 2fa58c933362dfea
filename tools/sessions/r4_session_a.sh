#!/bin/bash
# round 4, session a: first contact of the round's new code with the GPU -- smoke, the new tests (valence / collapsed hexes,
# launcher, probe, one-process, dying peer, two-base packed columns, golden fixtures), then the whole suite and a default bench line
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_a
mkdir -p $OUT
cd $R
python3 __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.txt
timeout 1500 python3 -m pytest tests/test_gpu_round4.py -m gpu -q -k "not 400_cubed" > $OUT/pytest_round4.txt 2>&1
echo "round4 rc=$?"; tail -15 $OUT/pytest_round4.txt | cut -c1-400
timeout 1200 python3 -m pytest tests/test_gpu_sharded.py -m gpu -q -k "launcher or probe or one_process or survivor" > $OUT/pytest_launcher.txt 2>&1
echo "launcher rc=$?"; tail -15 $OUT/pytest_launcher.txt | cut -c1-600
timeout 900 python3 bench.py --steps 3 --warmup 1 > $OUT/bench_default.json 2> $OUT/bench_err.txt
echo "bench rc=$?"; cut -c1-600 $OUT/bench_default.json
timeout 600 python3 tools/packed_ab.py 148,200 fp64 2 > $OUT/packed_ab.txt 2>&1; tail -8 $OUT/packed_ab.txt
timeout 3000 python3 -m pytest tests -m gpu -q -k "not 400_cubed" > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.txt | cut -c1-400
