#!/bin/bash
# round 3, session w: the driver's scaling command at full size as a dry run (8 and 2 rank processes SHARING the one GPU,
# shared-memory stand-in for RCCL, then peer to peer through HIP IPC), then the full GPU suite
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_w
mkdir -p $OUT
cd $R
export STAN_RCCL_LIB=$R/tests/fake_rccl/libfake_rccl.so STAN_BENCH_BACKEND=gloo STAN_BENCH_DEVICE=0
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    l = [x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1]
    d = json.loads(l)
    if d.get("value") is None: print(sys.argv[2], "ERROR LINE", l[:600]); sys.exit(0)
    c = d["config"]; ex = c.get("exchange", {})
    print(sys.argv[2], "DOF/s %.3e" % d["value"], "ms/step %.1f" % d["ms_per_step"], "its", c["cg_iterations"], "res %.6e" % c["rel_residual"], "converged", c["converged"],
          "| transport:", c["transport"][:70], "| per-rank frac %s" % (d["roofline"].get("per_rank", {}).get("frac")), "| allreduce us", ex.get("allreduce_us_per_call"), "halo us", ex.get("halo_us_per_call"))
except Exception as e:
    print(sys.argv[2], "FAILED", repr(e)); print(open(sys.argv[1]).read()[-1500:])
PY
}
for CFG in "8 " "8 --p2p" "2 " "2 --p2p"; do
  set -- $CFG
  tag=n$1$(echo $2 | tr -d '-')
  t0=$(date +%s)
  timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) bench.py --gpus $1 --steps 1 --warmup 1 --no-cpu $2 > $OUT/bench_148_$tag.txt 2> $OUT/bench_148_$tag.err
  echo "rc=$? wall $(( $(date +%s) - t0 )) s"
  line $OUT/bench_148_$tag.txt "148^3 --gpus $1 $2"
  tail -3 $OUT/bench_148_$tag.err | cut -c1-300
done
unset STAN_RCCL_LIB STAN_BENCH_BACKEND STAN_BENCH_DEVICE
timeout 600 python3 bench.py --steps 2 --warmup 1 --no-cpu > $OUT/bench_148_n1.txt 2>> $OUT/err.txt
line $OUT/bench_148_n1.txt "148^3 one rank"
timeout 3000 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.txt | cut -c1-300
