#!/bin/bash
# round 3, session f: IPC probe + diagnosis of the process-per-GPU peer-to-peer path, placement counter passes,
# stan_solver at 148^3 with the flat result writer, suite
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_f
mkdir -p $OUT
cd $R
timeout 120 tools/lab/ipc_probe > $OUT/ipc_probe.txt 2>&1; echo "ipc_probe rc=$?"; cat $OUT/ipc_probe.txt
for mode in 0 1; do
  for w in 3 4; do
    mkdir -p /tmp/ipc_$mode_$w
    STAN_P2P_WAIT_MODE=$mode STAN_RCCL_LIB=$R/tests/fake_rccl/libfake_rccl.so timeout 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) tests/sharded_worker.py 12 /tmp/ipc_$mode_$w 1 p2p > $OUT/ipc_p2p_mode${mode}_w${w}.txt 2>&1
    echo "wait_mode $mode world $w rc=$?"; grep "P2P rank" $OUT/ipc_p2p_mode${mode}_w${w}.txt | head -12; grep -i "error\|assert" $OUT/ipc_p2p_mode${mode}_w${w}.txt | head -5
  done
done
timeout 1500 python3 tools/cli_scale.py 148 > $OUT/cli_scale_n148.txt 2>&1
grep -E "^\{|wall" $OUT/cli_scale_n148.txt | cut -c1-1200
bash tools/placement_pmc.sh gpurun_out/r03_f/placement_pmc > $OUT/placement_pmc_log.txt 2>&1
cat $OUT/placement_pmc_log.txt | tail -12
ls -la $OUT/placement_pmc/TCC_EA0_RDREQ/*/ 2>/dev/null | head
find $OUT/placement_pmc -name "*.json" -size +20M -delete
du -sh $OUT/placement_pmc
timeout 3000 python3 -m pytest tests -m gpu -q -x --deselect tests/test_gpu_sharded.py::test_sharded_solve_peer_to_peer_between_processes > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -8 $OUT/pytest_gpu.txt | cut -c1-300
