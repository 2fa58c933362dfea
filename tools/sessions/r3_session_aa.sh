#!/bin/bash
# round 3, session aa: folded rows in sharded solves at real size (3 rank processes sharing the GPU; stand-in transport, then
# peer to peer through HIP IPC), against the one-rank line
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_aa
mkdir -p $OUT
cd $R
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    l = [x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1]
    d = json.loads(l)
    if d.get("value") is None: print(sys.argv[2], "ERROR LINE", l[:800]); sys.exit(0)
    c = d["config"]
    print(sys.argv[2], "DOF/s %.3e" % d["value"], "ms/step %.1f" % d["ms_per_step"], "its", c["cg_iterations"], "res %.6e" % c["rel_residual"], "converged", c["converged"], "streams", c["repacked_streams"], "| per-rank frac", d["roofline"].get("per_rank", {}).get("frac"))
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1]).read()[-1500:])
PY
}
timeout 600 python3 bench.py --size 120 --knockout 0.4 --steps 1 --warmup 1 --no-cpu > $OUT/one_rank.txt 2> $OUT/one_rank.err
line $OUT/one_rank.txt "one rank, fold auto"
export STAN_RCCL_LIB=$R/tests/fake_rccl/libfake_rccl.so STAN_BENCH_BACKEND=gloo STAN_BENCH_DEVICE=0
for CFG in "-1 " "0 " "-1 --p2p" "-1 --fixed48"; do
  set -- $CFG
  tag=fold$1$(echo $2 | tr -d '-')
  timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) bench.py --gpus 3 --size 120 --knockout 0.4 --fold $1 --steps 1 --warmup 1 --no-cpu $2 > $OUT/three_ranks_$tag.txt 2> $OUT/three_ranks_$tag.err
  echo "rc=$?"
  line $OUT/three_ranks_$tag.txt "3 ranks on one GPU, --fold $1 $2"
  grep -v "socket.cpp\|amdgpu.ids" $OUT/three_ranks_$tag.err | tail -3 | cut -c1-300
done
