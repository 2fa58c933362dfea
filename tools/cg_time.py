"""cg_ms and kernel split of one 148^3 solve x3 (for tools/lib_lab.sh A/B builds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch  # noqa
from stan_amd import hip, problem
job = problem.cube_job(148)
ctx = hip.Context(0); ctx.set_profiling(True); ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
for i in range(3):
    U, rep = K.cg_solve(job.F, 1e-8)
    p = ctx.profile()
    sp = p["spmv_ms_total"] + p["spmv2_ms_total"]
    print("solve %d: cg_ms %.1f  SpMV %.4f ms  non-SpMV per iteration %.4f ms  its %d" %
          (i, p["cg_ms"], p["spmv_ms_total"] / p["spmv_launches"], (p["cg_ms"] - sp) / rep["iterations"], rep["iterations"]), flush=True)
