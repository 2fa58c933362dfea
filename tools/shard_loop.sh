#!/bin/bash
# repeat a multi-rank sharded solve on one GPU through tests/fake_rccl (race hunting)
# usage: bash tools/shard_loop.sh <nranks> <spec> <repeats>
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/sd
for i in $(seq 1 $3); do
  STAN_RCCL_LIB=$GRAFT_REPO_ROOT/tests/fake_rccl/libfake_rccl.so timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $((29600+i)) tests/sharded_worker.py $2 /tmp/sd 1 > /tmp/sd/log_$i.txt 2>&1
  rc=$?
  echo "run $i rc=$rc"
  if [ $rc -ne 0 ]; then grep -E "fake_rccl|pair|Error|error" /tmp/sd/log_$i.txt | head -40; fi
done
