"""Does the SpMV time depend on where hipMalloc puts the matrix?  Re-allocate (pool off) in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
job = problem.cube_job(n)
ctx = hip.Context(0)
ctx.set_option(hip.OPT_POOL, 0)
hold = []
for i in range(8):
    K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    t = [K.spmv_bench(20) for _ in range(3)]
    print("allocation %d: spmv_bench %.4f %.4f %.4f ms" % (i, *t), flush=True)
    K.free()
    if i % 2 == 1:   # shift the next placement: keep an odd-sized block alive
        hold.append(torch.empty((1 << 27) + 12345 * (i + 1), dtype=torch.uint8, device="cuda"))
