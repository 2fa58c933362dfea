"""Per-iteration time of the CG at small sizes (is the loop launch-bound there?  No: a hipGraph replay of
a captured 10-iteration chunk was built in round 2 and measured the same 28-33 us per iteration up to
47 k DOF -- profiles/r02/small_sizes_graph_replay_vs_eager.txt -- the three kernels of an iteration are
chains of dependent memory round trips, ~10 us each; the replay code was removed again).
usage: small_sizes.py [sizes=8,16,24,40,56]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from stan_amd import hip, problem
sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "8,16,24,40,56").split(",")]
ctx = hip.Context(0)
import time
ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
for n in sizes:
    job = problem.cube_job(n)
    K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    d_F = torch.from_numpy(job.F).cuda(); d_U = torch.zeros_like(d_F)
    line = "n %3d  %8d DOF:" % (n, job.n_dof)
    for tag, g in (("workgroup per slice", 1 << 40), ("wavefront per slice", 0)):
        ctx.set_option(hip.OPT_SPMV_SMALL, g)
        best = 1e9
        for rep_ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rep = K.cg_solve_dev(d_F.data_ptr(), d_U.data_ptr(), 1e-8)
            best = min(best, time.perf_counter() - t0)
        line += "  %s %.3f ms = %.2f us per iteration (%d its)" % (tag, 1e3 * best, 1e6 * best / max(rep["iterations"], 1), rep["iterations"])
    print(line, flush=True)
    K.free()
ctx.set_option(hip.OPT_SPMV_SMALL, 1)
ctx.close()
