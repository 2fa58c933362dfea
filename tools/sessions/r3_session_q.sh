#!/bin/bash
# round 3, session q: wait mode 2 (reductions polled inside the consuming kernel), p2p tests, repetitions
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_q
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_round3.py -m gpu -q -k "peer_to_peer" > $OUT/pytest_p2p.txt 2>&1
echo "p2p tests rc=$?"; tail -6 $OUT/pytest_p2p.txt | cut -c1-300
export STAN_RCCL_LIB=$R/tests/fake_rccl/libfake_rccl.so GPU_MAX_HW_QUEUES=12 STAN_DEBUG_STALL_S=15 STAN_P2P_WAIT_MODE=2
hang=0; tot=0
for rep in 1 2 3; do
for cfg in "10 3 0 1" "10 3 1 1" "12 3 1 1" "10 2 1 1" "14 4 1 1" "9 4 0 1"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout 60 python3 tools/p2p_hang.py $cfg > $OUT/mode2_${tag}_$rep.txt 2>&1
  rc=$?; tot=$((tot+1)); if [ $rc -ne 0 ]; then hang=$((hang+1)); echo "mode 2 cfg [$cfg] rep $rep rc=$rc"; grep -v amdgpu.ids $OUT/mode2_${tag}_$rep.txt | head -20; fi
done
done
echo "WAIT MODE 2: $hang stalls of $tot runs"
timeout 600 python3 tools/p2p_latency.py 48 2 > $OUT/p2p_latency_n48_2ranks_mode2.jsonl 2>> $OUT/err.txt
cut -c1-330 $OUT/p2p_latency_n48_2ranks_mode2.jsonl | head -8
unset STAN_P2P_WAIT_MODE
timeout 3000 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.txt | cut -c1-300
