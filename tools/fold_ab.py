"""A/B in ONE process on ONE matrix: folded reductions on/off, single-reduction loop, and the SpMV
back to back -- in-CG SpMV ms and whole-solve ms.  usage: fold_ab.py [n=148] [rounds=3]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
from stan_amd import hip, problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 148
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
job = problem.cube_job(n)
ctx = hip.Context(0)
ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
ctx.set_option(hip.OPT_PLACEMENT_TRIES, 24)
ctx.set_profiling(True)
K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
pr = ctx.profile()
print("placement search: %d candidates, probe kept %.4f ms, slowest %.4f ms" %
      (pr["placement_candidates"], pr["placement_ms_best"], pr["placement_ms_worst"]))
K.cg_solve(job.F, 1e-8)   # scales the matrix, warms up
print("back to back (spmv_bench, own buffers): %.4f ms" % K.spmv_bench(20))
for r in range(rounds):
    for tag, fold, sr in (("fold", 1, 0), ("no fold", 0, 0), ("single-reduce", 1, 1)):
        ctx.set_option(hip.OPT_CG_FOLD_REDUCE, fold)
        ctx.set_option(hip.OPT_CG_SINGLE_REDUCE, sr)
        U, rep = K.cg_solve(job.F, 1e-8)
        p = ctx.profile()
        print("round %d %-14s its %d  cg %.1f ms  in-CG SpMV %.4f ms (%d launches)  2-product %.4f ms  launches/it %.2f" %
              (r, tag, rep["iterations"], p["cg_ms"], p["spmv_ms_total"] / max(p["spmv_launches"], 1), p["spmv_launches"],
               p["spmv2_ms_total"] / max(p["spmv2_launches"], 1), p["loop_kernel_launches"] / max(p["loop_iterations_enqueued"], 1)))
ctx.set_option(hip.OPT_CG_FOLD_REDUCE, 1); ctx.set_option(hip.OPT_CG_SINGLE_REDUCE, 0)
for r in range(rounds):
    for v in ((9, 17, 18, 12, 0) if hasattr(ctx.lib, "stan_hip_lab_incg_penalty") else (9, 12, 0)):
        ctx.set_option(hip.OPT_SPMV_VARIANT, v)
        U, rep = K.cg_solve(job.F, 1e-8)
        p = ctx.profile()
        print("round %d kernel variant %2d: cg %.1f ms  in-CG SpMV %.4f ms  2-product %.4f ms" %
              (r, v, p["cg_ms"], p["spmv_ms_total"] / max(p["spmv_launches"], 1), p["spmv2_ms_total"] / max(p["spmv2_launches"], 1)))
ctx.set_option(hip.OPT_SPMV_VARIANT, -1)
for r in range(rounds):
    for pol in (0, 1, 2, 3):
        ctx.set_option(hip.OPT_VEC_STORE_NT, pol)
        U, rep = K.cg_solve(job.F, 1e-8)
        p = ctx.profile()
        sp = p["spmv_ms_total"] + p["spmv2_ms_total"]
        print("round %d vector store policy %d (bit0 p nt, bit1 r nt): cg %.1f ms  in-CG SpMV %.4f ms  non-SpMV per iteration %.4f ms" %
              (r, pol, p["cg_ms"], p["spmv_ms_total"] / max(p["spmv_launches"], 1), (p["cg_ms"] - sp) / rep["iterations"]))
ctx.set_option(hip.OPT_VEC_STORE_NT, 3)
for r in range(rounds):
    for d in (0, 1):
        ctx.set_option(hip.OPT_CG_DEFER_X, d)
        U, rep = K.cg_solve(job.F, 1e-8)
        p = ctx.profile()
        sp = p["spmv_ms_total"] + p["spmv2_ms_total"]
        print("round %d deferred x update %d: cg %.1f ms  in-CG SpMV %.4f ms  non-SpMV per iteration %.4f ms" %
              (r, d, p["cg_ms"], p["spmv_ms_total"] / max(p["spmv_launches"], 1), (p["cg_ms"] - sp) / rep["iterations"]))
ctx.set_option(hip.OPT_CG_DEFER_X, 1)
if hasattr(ctx.lib, "stan_hip_lab_incg_penalty"):   # lab build only
    import ctypes as C
    out = np.zeros(15)
    ctx._chk(ctx.lib.stan_hip_lab_incg_penalty(ctx.h, K.k, C.c_int32(30), out.ctypes.data_as(C.POINTER(C.c_double))))
    print("SpMV alone, events around each product: back to back %.4f | gather vector rewritten before each %.4f | "
          "rewritten + a k_step pass in between %.4f | only the k_step pass %.4f | rewritten with nt stores %.4f | "
          "rewritten, then read once by a streaming kernel %.4f | nt stores, then read once %.4f | rewritten with agent-scope (sc1) stores %.4f | "
          "with system-scope stores %.4f | read-modify-write like k_update: plain load + plain store %.4f | plain load + nt store %.4f | "
          "nt load + nt store %.4f | nt load + plain store %.4f | ping-pong (written to the other of two buffers): nt load + nt store %.4f | "
          "plain load + nt store %.4f ms" % tuple(out))
K.free(); ctx.close()
