import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from stan_amd import hip, problem
job = problem.cube_job(40)
args = (job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
f = lambda: torch.cuda.mem_get_info(0)[0] / 1e6
f0 = f(); print("start free %.0f MB" % f0)
ctx = hip.Context(0); print("after ctx: used %.0f MB" % (f0 - f()))
for i in range(2):
    K = ctx.assemble_hex8(*args); print(" assembled: used %.0f MB" % (f0 - f()))
    U, rep = K.cg_solve(job.F, 1e-8); print(" solved: used %.0f MB" % (f0 - f()))
    K.free(); print(" freed: used %.0f MB" % (f0 - f()))
ctx.set_option(hip.OPT_POOL, 0); print("pool off: used %.0f MB" % (f0 - f()))
K = ctx.assemble_hex8(*args); print(" assembled: used %.0f MB" % (f0 - f())); K.free(); print(" freed: used %.0f MB" % (f0 - f()))
ctx.close(); print("closed: used %.0f MB" % (f0 - f()))
