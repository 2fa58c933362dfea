"""Stage-by-stage GPU bring-up check (prints instead of asserting)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import pyoracle as O
from stan_amd import hip, problem
from tests.util import random_hexes

ctx = hip.Context(0)
xs = random_hexes(8, seed=1)
for t in (1, 2):
    Kg = ctx.ke_hex8_batch(xs, 7e4, 0.33, np.full(8, t, np.uint8))
    err = max(np.abs(Kg[i] - O.ke_hex8(xs[i], 7e4, 0.33, t)[1]).max() / np.abs(Kg[i]).max() for i in range(8))
    print("ke type", t, "relerr", err)
for n in (1, 2, 3, 8):
    job = problem.cube_job(n, jitter=0.1)
    K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    print("n", n, K.info())
    rp, col, val = K.to_csr()
    rc, A = O.assemble(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    print("  pattern", np.array_equal(rp, A.ridx), np.array_equal(col, A.idx), "nnz", rp[-1], A.nnz)
    if rp[-1] == A.nnz:
        print("  valerr", np.abs(val - A.vals).max() / np.abs(A.vals).max())
    x = np.random.default_rng(0).standard_normal(job.n_red)
    y = K.spmv(x); yo = O.smv_upper(A, x)
    print("  spmv err", np.abs(y - yo).max() / np.abs(yo).max())
    U, rep = K.cg_solve(job.F, 1e-12)
    Uo, repo = O.cg(A, job.F, 1e-12)
    print("  cg", rep, repo, "uerr", np.abs(U - Uo).max() / np.abs(Uo).max())
    K.free()
for n in (40, 100):
    t0 = time.time(); job = problem.cube_job(n); t1 = time.time()
    ctx.set_profiling(True)
    K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    t2 = time.time()
    U, rep = K.cg_solve(job.F, 1e-8)
    t3 = time.time()
    print("n", n, "job", t1 - t0, "asm(h2d incl)", t2 - t1, "cg", t3 - t2, rep, ctx.profile())
    ms = K.spmv_bench(20)
    info = K.info(); p = ctx.profile()
    print("  spmv ms", ms, "GB/s", p["spmv_bytes"] / ms / 1e6, info)
    K.free()
ctx.close()
