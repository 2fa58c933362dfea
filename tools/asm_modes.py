"""Assembly mode 0 (row-owner gather) vs mode 1 (element wave + colour-ordered scatter) at the
bench size, in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
job = problem.cube_job(n)
ctx = hip.Context(0); ctx.set_profiling(True)
vals = {}
for rnd in range(3):
    for mode in (0, 1):
        ctx.set_option(hip.OPT_ASSEMBLY_MODE, mode)
        K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
        p = ctx.profile()
        print("round %d mode %d: symbolic %.2f ms numeric %.2f ms colours %d" % (rnd, mode, p["symbolic_ms"], p["numeric_ms"], p["assembly_colours"]), flush=True)
        if rnd == 0 and n <= 40:
            vals[mode] = K.to_csr()[2]
        K.free()
if vals:
    print("max rel diff mode0 vs mode1:", np.abs(vals[0] - vals[1]).max() / np.abs(vals[0]).max())
