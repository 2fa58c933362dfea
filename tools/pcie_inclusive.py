"""Times the host-pointer entry points (H2D + assembly, H2D + CG + D2H) at the bench size:
the PCIe-inclusive rate DESIGN.md quotes next to bench.py's HBM-resident value."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
job = problem.cube_job(n)
ctx = hip.Context(0)
ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
for rep in range(2):
    t0 = time.perf_counter()
    K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    t1 = time.perf_counter()
    U, r = K.cg_solve(job.F, 1e-8)
    t2 = time.perf_counter()
    K.free()
    print("n=%d host-pointer path: assemble %.1f ms, cg %.1f ms, total %.1f ms -> %.3f M DOF/s (its %d type %d)" %
          (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, job.n_dof / (t2 - t0) / 1e6, r["iterations"], r["terminationtype"]))
