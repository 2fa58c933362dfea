"""In-CG SpMV time (working launches, HIP events) per kernel variant and value stream, one process.
usage: python tools/incg_variants.py [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
job = problem.cube_job(n)
ctx = hip.Context(0)
ctx.set_profiling(True)
ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
for prec, name in ((hip.PREC_FP64, "fp64"), (hip.PREC_FIXED48, "fixed48")):
    for rnd in range(2):
        for v in (0, 1, 9, 5, 12):
            ctx.set_option(hip.OPT_SPMV_VARIANT, v)
            U, rep = K.cg_solve(job.F, 1e-8, precision_mode=prec)
            p = ctx.profile()
            print("%-8s variant %2d: cg_ms %.1f  in-CG SpMV %.4f ms (%d launches)  two-product %.4f ms  its %d" %
                  (name, v, p["cg_ms"], p["spmv_ms_total"] / p["spmv_launches"], p["spmv_launches"],
                   p["spmv2_ms_total"] / max(p["spmv2_launches"], 1), rep["iterations"]), flush=True)
