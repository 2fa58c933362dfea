"""What an exchange of the sharded CG costs the stream, RCCL stand-in vs peer to peer, with the ranks of ONE
process sharing GPU 0 (the only multi-rank topology a one-GPU box offers: the kernels of the ranks time-slice
the device, so the absolute iteration times say little; the per-exchange stream times and the launch counts do).
  STAN_RCCL_LIB=tests/fake_rccl/libfake_rccl.so GPU_MAX_HW_QUEUES=12 python tools/p2p_latency.py [n] [nranks]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
nranks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
job = problem.cube_job(n)
ctx = hip.Context(devices=[0] * nranks)
ctx.set_profiling(True)
ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
for sr in (0, 1):
    ctx.set_option(hip.OPT_CG_SINGLE_REDUCE, sr)
    for p2p in (0, 1, 0, 1):
        ctx.set_option(hip.OPT_COMM_P2P, p2p)
        U, rep = K.cg_solve(job.F, 1e-8)
        pr = ctx.profile()
        its = max(pr["loop_iterations_enqueued"], 1)
        print(json.dumps({"n": n, "ranks_on_gpu0": nranks, "loop": "single-reduction" if sr else "classic",
                          "transport": "peer to peer" if p2p else "RCCL stand-in (host-staged)",
                          "iterations": rep["iterations"], "cg_ms": pr["cg_ms"], "ms_per_iteration": pr["cg_ms"] / max(rep["iterations"], 1),
                          "kernels_per_iteration": pr["loop_kernel_launches"] / its,
                          "collectives_per_iteration": pr["loop_collectives"] / its,
                          "stream_waits_per_iteration": pr["loop_stream_waits"] / its,
                          "reduce_us_per_exchange_rank0": 1e3 * pr["comm_reduce_ms_total"] / max(pr["comm_reduce_calls"], 1),
                          "halo_us_per_exchange_rank0": 1e3 * pr["comm_halo_ms_total"] / max(pr["comm_halo_calls"], 1)}), flush=True)
K.free(); ctx.close()
