export STAN_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so STAN_BENCH_BACKEND=gloo STAN_BENCH_DEVICE=0
for mode in sync async asynchost; do
  export FAKE_RCCL_ASYNC=0; unset FAKE_RCCL_ASYNC_HOST_BOXES
  [ $mode = async ] && export FAKE_RCCL_ASYNC=1
  [ $mode = asynchost ] && export FAKE_RCCL_ASYNC=1 FAKE_RCCL_ASYNC_HOST_BOXES=1
  timeout 300 python bench.py --gpus 2 --steps 1 --warmup 1 --size 100 --no-cpu --no-p2p-probe > gpurun_out/r06_exch_$mode.json 2> gpurun_out/r06_exch_$mode.err
  grep -h "fake_rccl:" gpurun_out/r06_exch_$mode.err | head -2
  python - <<PY
import json
d=json.loads(open("gpurun_out/r06_exch_$mode.json").read().strip().splitlines()[-1])
e=d["config"]["exchange"]
print("$mode", "value %.3e" % d["value"], "ms/step %.1f" % d["ms_per_step"], "halo us", [round(x,1) for x in e["halo_us_per_call"]], "allreduce us", [round(x,1) for x in e["allreduce_us_per_call"]], "spmv ms", d["roofline"]["per_rank"]["avg_launch_ms"], d["config"]["cg_iterations"])
PY
done
