"""Allocation by trial (STAN_OPT_PLACEMENT_TRIES): distribution of the SpMV time over fresh allocations
with 1 try and with 4 tries, one process, pool off so that every assembly allocates anew."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
job = problem.cube_job(n)
ctx = hip.Context(0)
ctx.set_option(hip.OPT_POOL, 0)
for tries in (1, 4, 1, 4):
    ctx.set_option(hip.OPT_PLACEMENT_TRIES, tries)
    ts = []
    for i in range(6):
        K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
        ts.append(min(K.spmv_bench(20) for _ in range(2)))
        K.free()
    print("tries %d: spmv_bench over 6 fresh allocations: %s  (median %.4f)" % (tries, " ".join("%.4f" % t for t in ts), np.median(ts)), flush=True)
