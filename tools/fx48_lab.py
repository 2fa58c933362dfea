"""fp64 vs FIXED-48 matrix stream in ONE process: SpMV time (interleaved rounds), CG to 1e-8
(iterations, wall time), deviation of U.  usage: python tools/fx48_lab.py [n]"""
import sys, os, time
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
info = K.info()
nb, nloc = info["n_blocks"], info["n_block_rows"]
modes = {"fp64": (hip.PREC_FP64, 76), "fixed48": (hip.PREC_FIXED48, 60), "fp32": (hip.PREC_MIXED, 40)}
sol = {}
for name, (prec, bpb) in modes.items():
    t0 = time.perf_counter()
    U, rep = K.cg_solve(job.F, 1e-8, precision_mode=prec)
    dt = time.perf_counter() - t0
    p = ctx.profile()
    sol[name] = U
    ms = p["spmv_ms_total"] / max(p["spmv_launches"], 1)
    print("%-8s CG: type %d, %d iterations, cg_ms %.1f (wall %.2f s incl. copies), SpMV %.4f ms for %.3f GB "
          "-> %.0f GB/s (%.1f%% of 8 TB/s), value_stream %d" %
          (name, rep["terminationtype"], rep["iterations"], p["cg_ms"], dt, ms, p["spmv_bytes"] / 1e9,
           p["spmv_bytes"] / ms / 1e6, p["spmv_bytes"] / ms / 1e6 / 80, p["value_stream"]))
for name in ("fixed48", "fp32"):
    print("%-8s max |U - U_fp64| / max |U_fp64| = %.2e" %
          (name, np.abs(sol[name] - sol["fp64"]).max() / np.abs(sol["fp64"]).max()))
res = {k: [] for k in modes}
for rnd in range(5):
    for name, (prec, bpb) in modes.items():
        res[name].append(K.spmv_bench(20, prec))
for name, (prec, bpb) in modes.items():
    t = np.array(res[name]); b = nb * bpb + nloc * 52
    print("%-8s spmv_bench: median %.4f ms min %.4f ms, %.3f GB -> %.0f GB/s (%.1f%% of 8 TB/s)" %
          (name, np.median(t), t.min(), b / 1e9, b / np.median(t) / 1e6, b / np.median(t) / 1e6 / 80))
for v in (0, 1, 4, 5, 3):
    ctx.set_option(hip.OPT_SPMV_VARIANT, v)
    t = [K.spmv_bench(20, hip.PREC_FIXED48) for _ in range(3)]
    print("fixed48 variant %d (bit0 non-temporal loads, bit1 XCD-contiguous slices, bit2 unroll 4): median %.4f ms" % (v, np.median(t)))
