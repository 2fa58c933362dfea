"""BASELINE.json config 4 at its own size and rank count: the n^3 cube row-partitioned over
`nranks` ranks of ONE process (stan_hip_init_multi -- what Solver.cs:18-69's single process would use), bench mode
(merit stop off, eps from the fixture).  On the one-GPU test box every rank drives GPU 0 and RCCL is tests/fake_rccl
(real RCCL refuses two ranks on one device); sharding, halo plan, the exchanges of the loop and the result segments
are the product.
usage: config4_worker.py <n> <nranks> <rccl|p2p> <golden.npz> <out.npz>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from stan_amd import hip, problem  # noqa: E402

n, nranks, transport, golden, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
g = np.load(golden)
t0 = time.time()
job = problem.cube_job(n)
t_job = time.time() - t0
ctx = hip.Context(devices=[0] * nranks)
ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
if transport == "p2p":
    ctx.set_option(hip.OPT_COMM_P2P, 1)
    assert ctx.comm_info()["p2p"]
ctx.set_profiling(True)
t0 = time.time()
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
t_asm = time.time() - t0
parts = [K.part_info(r) for r in range(nranks)]
t0 = time.time()
U, rep = K.cg_solve(job.F, float(g["eps"]))
t_cg = time.time() - t0
profs = [ctx.profile_rank(r) for r in range(nranks)]
print("config 4 worker: %d^3 on %d ranks (%s): mesh %.1f s, assemble %.1f s, solve %.1f s, %s" %
      (n, nranks, transport, t_job, t_asm, t_cg, rep), flush=True)
np.savez(out, U_at_idx=U[g["idx"]], u_max=np.abs(U).max(), u_sum=U.sum(), u_l2=np.sqrt(U @ U),
         its=rep["iterations"], term=rep["terminationtype"], rel=rep["rel_residual"],
         its_rank=np.array([p["iterations"] for p in profs]), term_rank=np.array([p["termination_type"] for p in profs]),
         coll=np.array([p["loop_collectives"] for p in profs]), waits=np.array([p["loop_stream_waits"] for p in profs]),
         halo_rows=np.array([p["n_halo"] for p in parts]), row_begin=np.array([p["row_begin"] for p in parts]),
         row_end=np.array([p["row_end"] for p in parts]), n_blocks=np.array([p["n_blocks"] for p in parts]),
         seconds=np.array([t_job, t_asm, t_cg]))
K.free()
ctx.close()
