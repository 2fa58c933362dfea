#!/usr/bin/env python3
"""Reduced-precision value streams against the fp64 solve, at a cube size, bench mode (merit stop off, eps 1e-8):
what each STAN_OPT_CG_REFINE setting delivers (fp64 residual as the library reports it, checked here against an
independent product on the unscaled matrix) and what it costs.  One JSON line per leg.
usage: mixed_refine.py [n=148] [eps=1e-8]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from stan_amd import hip, problem  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
job = problem.cube_job(n)
dev = torch.device("cuda", 0)
ctx = hip.Context(0)
ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
ctx.set_option(hip.OPT_POOL_MAX_BYTES, int(0.9 * torch.cuda.mem_get_info(dev)[0]))
ctx.set_profiling(True)
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
g = None
gp = os.path.join(ROOT, "tests", "golden", "bench_mode_%d.npz" % n)
if os.path.exists(gp):
    g = np.load(gp)
fn = np.linalg.norm(job.F)
legs = [("fp64", hip.PREC_FP64, 1, 10), ("mixed refine 0 (check only)", hip.PREC_MIXED, 0, 10),
        ("mixed refine 1 (passes)", hip.PREC_MIXED, 1, 10), ("mixed refine 2 (fp64 refresh every 10)", hip.PREC_MIXED, 2, 10),
        ("mixed refine 2 (fp64 refresh every 20)", hip.PREC_MIXED, 2, 20), ("mixed refine 2 (fp64 refresh every 50)", hip.PREC_MIXED, 2, 50),
        ("fixed48 refine 1", hip.PREC_FIXED48, 1, 10)]
for name, prec, refine, rup in legs:
    ctx.set_option(hip.OPT_CG_REFINE, refine)
    ctx.set_option(hip.OPT_CG_RUPDATE, rup)
    for rep_i in range(2):            # the second run is the timed one (value-stream copies exist by then)
        t0 = time.perf_counter()
        U, rep = K.cg_solve(job.F, eps, 0, prec)
        dt = time.perf_counter() - t0
    pr = ctx.profile()
    out = {"leg": name, "n": n, "eps": eps, "report": rep, "cg_ms": pr["cg_ms"], "host_wall_ms": dt * 1e3,
           "passes": pr["refine_passes"], "rel_recurrence": pr["rel_residual_recurrence"], "rel_fp64_library": pr["rel_residual_fp64"],
           "fp64_products": pr["fp64_products"], "fp64_products_ms": pr["fp64_products_ms"],
           "spmv_avg_ms": pr["spmv_ms_total"] / max(pr["spmv_launches"], 1), "spmv_launches": pr["spmv_launches"],
           "spmv2_launches": pr["spmv2_launches"]}
    out["rel_fp64_independent"] = K.scaled_residual(job.F, U)   # stan_hip_spmv + exported diagonal, numpy
    if g is not None:
        out["max_dU_over_max_U_vs_oracle"] = float(np.abs(U[g["idx"]] - g["U"]).max() / float(g["u_max"]))
        out["oracle_iterations"] = int(g["iterations"])
    print(json.dumps(out), flush=True)
ctx.set_option(hip.OPT_CG_RUPDATE, 10)
K.free()
ctx.close()
