import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa
from stan_amd import hip, problem
n = int(os.environ.get("LAB_N", "148"))
job = problem.cube_job(n)
ctx = hip.Context(0); ctx.set_profiling(True)
ts = []
for i in range(3):
    K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    p = ctx.profile(); ts.append((p["symbolic_ms"], p["numeric_ms"]))
    if i == 2:
        rp, col, val = (None, None, None)
    K.free()
print("symbolic %.2f ms numeric %.2f ms (min of 3)" % (min(t[0] for t in ts), min(t[1] for t in ts)))
