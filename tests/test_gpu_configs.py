"""BASELINE.json's configurations at their own sizes on one GPU: config 3 (200^3, properties the oracle cannot check in
seconds), config 5 in its two well-posed halves (400^3 HEX8_G1 assembly; 400^3 fp32 matrix refined to 1e-8 in fp64
terms) and its combination at a size the oracle can check.  Config 4 (200^3 on 8 ranks) lives with the other
one-process multi-rank tests in test_gpu_multi.py; configs 2 / 3 against the oracle's fixtures in test_gpu_cg_loops.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from stan_amd import problem
from stan_amd.cube import cube_mesh, revolved_mesh

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")
U_TOL = 1e-6
K_TOL = 1e-13
OPT_ASSEMBLY_MODE = 5
OPT_FOLD = 19
OPT_SELL_SIGMA, OPT_MERIT = 17, 1


def _assemble_both(ctx, oracle, job):
    K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                          job.mat_E_nu, job.red)
    rc, A = oracle.assemble(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                            job.mat_E_nu, job.red, n_threads=8)
    assert rc == 0
    return K, A


def test_config3_200_cubed_properties(gpu_ctx):
    """BASELINE.json config 3 (200^3, 24.36 M DOF, fp64, the HBM-roofline run): the oracle cannot
    run it in seconds, so size-independent properties -- SURVEY.md section 8's counts, the int32
    slot guard, symmetry of the operator, CG to 1e-8 (merit stop off: bench mode) checked by an
    independent product with a freshly assembled, unscaled K, and the folded / separate reductions
    giving the same bits at this size."""
    from stan_amd import hip
    n = 200
    job = problem.cube_job(n)
    assert (job.n_dof, job.n_red) == (24361803, 24240600)
    K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                              job.mat_E_nu, job.red)
    info = K.info()
    assert info["n_blocks"] == (3 * n + 1) ** 3 == 217081801 and info["max_row_blocks"] == 27
    assert info["n_block_rows"] == (n + 1) ** 3
    # slots are int32 in the layout, byte offsets 64-bit: 200^3 needs 219.6 M slots-of-64 ...
    assert info["n_blocks"] <= info["n_slots"] * 64 <= 1.02 * info["n_blocks"]
    assert info["n_slots"] < 2 ** 31 and info["n_slots"] * 64 * 72 > 2 ** 32   # ... and > 4 GiB of values
    rng = np.random.default_rng(11)
    x, y = rng.standard_normal(job.n_red), rng.standard_normal(job.n_red)
    Kx, Ky = K.spmv(x), K.spmv(y)
    assert abs(y @ Kx - x @ Ky) <= 1e-9 * abs(y @ Kx)
    gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
    try:
        U, rep = K.cg_solve(job.F, 1e-8)
        gpu_ctx.set_option(hip.OPT_CG_FOLD_REDUCE, 0)
        U2, rep2 = K.cg_solve(job.F, 1e-8)
    finally:
        gpu_ctx.set_option(hip.OPT_CG_FOLD_REDUCE, 1)
        gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 1)
    assert rep["terminationtype"] == 1 and rep["rel_residual"] <= 1e-8
    assert rep == rep2 and np.array_equal(U, U2)
    K.free()
    Kf = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                               job.mat_E_nu, job.red)
    r = job.F - Kf.spmv(U)
    print("200^3: %d iterations, independent residual %.2e" % (rep["iterations"], np.linalg.norm(r) / np.linalg.norm(job.F)))
    assert np.linalg.norm(r) <= 1e-6 * np.linalg.norm(job.F)
    disp = np.zeros(job.n_dof); disp[job.red != -1] = U
    uz = disp[job.node_dof[:, 2]]
    assert uz.min() > -1e-9 * uz.max() and uz.argmax() in np.nonzero(job.xyz[:, 0] == n)[0]
    Kf.free()


def test_config5_g1_mixed_precision_three_face_clamp(gpu_ctx, oracle):
    """BASELINE.json config 5's combination at a size the oracle can check: HEX8_G1
    (FE_Library.cs:63-89) + STAN_PREC_MIXED (fp32 matrix, fp64 vectors), clamps on x=0, y=0 and
    z=0 (with only x=0 clamped the G1 operator is singular: SURVEY.md App. D).  n = 16:
    kappa(S K S) = 1.55e5, so rounding the scaled entries to fp32 (relative 6e-8) may move the
    solution by up to kappa * 6e-8 = 9.3e-3; direct solves of both matrices in the build
    container differ by 1.5e-4."""
    from stan_amd import hip
    n = 16
    kappa = 1.55e5
    job = problem.cube_job(n, etype=1)          # clamp_faces = "xyz" for G1
    assert job.n_fixed == 3 * ((n + 1) ** 3 - n ** 3)
    K, A = _assemble_both(gpu_ctx, oracle, job)
    rowptr, col, val = K.to_csr(upper_only=True)
    assert np.array_equal(rowptr, A.ridx) and np.array_equal(col, A.idx)
    assert np.abs(val - A.vals).max() <= 1e-13 * np.abs(A.vals).max()
    gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
    gpu_ctx.set_profiling(True)
    try:
        Um, rm = K.cg_solve(job.F, 1e-8, precision_mode=hip.PREC_MIXED)
        assert gpu_ctx.profile()["value_stream"] == hip.PREC_MIXED
        U64, r64 = K.cg_solve(job.F, 1e-8)
        Uo, ro = oracle.cg(A, job.F, 1e-8, merit_stop=False)
    finally:
        gpu_ctx.set_profiling(False)
        gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 1)
    assert rm["terminationtype"] == r64["terminationtype"] == ro["terminationtype"] == 1
    assert abs(r64["iterations"] - ro["iterations"]) <= max(3, ro["iterations"] // 50)
    # round 5: the fp32 copy is refined until the FP64 residual meets 1e-8 (STAN_OPT_CG_REFINE): more iterations than the
    # fp64 stream, and an answer of its quality instead of one that is off by kappa * 6e-8
    assert ro["iterations"] <= rm["iterations"] <= 3 * ro["iterations"] and rm["rel_residual"] <= 1e-8
    d64 = np.abs(U64 - Uo).max() / np.abs(Uo).max()
    dm = np.abs(Um - Uo).max() / np.abs(Uo).max()
    print("G1 + mixed at 16^3: its %d (fp64 %d, oracle %d), |U-Uo| mixed %.2e, fp64 %.2e"
          % (rm["iterations"], r64["iterations"], ro["iterations"], dm, d64))
    assert d64 <= kappa * 1e-8
    assert dm <= kappa * 1e-8
    K.free()


def test_config5_at_size_400_cubed(gpu_ctx, oracle):
    """BASELINE config 5 at its size (193 M DOF on ONE GPU).  The combination as named -- G1 elements AND a 1e-8 solve --
    is ill-posed at this size (hourglass modes: profiles/r02/CONFIG5.md), so the two halves are checked where each is
    well-posed:
      G1  400^3 HEX8_G1 assembly: block count, symmetry, the rigid-translation null vector on rows away from the clamp,
          and the three columns of an interior node against the ORACLE's columns of the same stencil (a 6^3 cube: an
          interior row of a uniform mesh does not depend on the mesh size);
      G2  400^3 HEX8_G2, fp32 matrix / fp64 vectors (STAN_PREC_MIXED) to 1e-8 IN FP64 TERMS (refinement passes), with an
          INDEPENDENT residual: F - K U through the library's plain fp64 product on the unscaled matrix."""
    import time
    from stan_amd import hip
    avail_gb = 0
    for ln in open("/proc/meminfo"):
        if ln.startswith("MemAvailable"):
            avail_gb = int(ln.split()[1]) / 1e6
    if avail_gb < 48:
        pytest.skip("host has %.0f GB available: the 400^3 mesh arrays need ~30 GB" % avail_gb)
    free_b = __import__("torch").cuda.mem_get_info(0)[0]
    if free_b < 230e9:
        pytest.skip("GPU has %.0f GB free: 400^3 needs ~215 GB (fp64 values + fp32 copy)" % (free_b / 1e9))
    t0 = time.time()
    n = 400
    # ---- G1 assembly properties
    job = problem.cube_job(n, etype=problem.HEX8_G1)           # clamp x = 0, y = 0, z = 0
    t_mesh = time.time() - t0
    K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    info = K.info()
    assert info["n_blocks"] == (3 * n + 1) ** 3 and info["n_dof"] == 3 * (n + 1) ** 3 == 193_443_603
    rng = np.random.default_rng(400)
    x, y = rng.standard_normal(job.n_red), rng.standard_normal(job.n_red)
    Kx, Ky = K.spmv(x), K.spmv(y)
    assert abs(y @ Kx - x @ Ky) <= 1e-10 * (np.abs(y) @ np.abs(Kx))                       # symmetry
    # rigid translation in x: K t = 0 on every free DOF whose node couples to no clamped node (i, j, k >= 2)
    m = n + 1
    t = np.zeros(job.n_dof)
    t[0::3] = 1.0
    full_to_red = np.nonzero(job.red != -1)[0]
    Kt = K.spmv(t[full_to_red])
    idx = np.arange(m ** 3)
    far = ((idx % m) >= 2) & (((idx // m) % m) >= 2) & ((idx // (m * m)) >= 2)
    dof_far = (job.node_dof.reshape(-1, 3)[far]).ravel()
    red_far = dof_far - job.red[dof_far]
    scale = np.abs(Kx).max() / np.abs(x).max()
    assert np.abs(Kt[red_far]).max() <= 1e-9 * scale
    # an interior node's three columns against the oracle's (same uniform stencil in a 6^3 cube)
    small = problem.cube_job(6, etype=problem.HEX8_G1)
    rc, A = oracle.assemble(small.xyz, small.node_dof, small.conn, small.elem_mat, small.elem_type, small.mat_E_nu, small.red)
    As = A.to_scipy_full().tocsc()

    def node(nn, i, j, k):
        return i + (nn + 1) * (j + (nn + 1) * k)
    c_big, c_small = node(n, 200, 200, 200), node(6, 3, 3, 3)
    for d in range(3):
        e = np.zeros(job.n_red)
        gd = job.node_dof.reshape(-1, 3)[c_big, d]
        e[gd - job.red[gd]] = 1.0
        col = K.spmv(e)
        sd = small.node_dof.reshape(-1, 3)[c_small, d]
        ocol = np.asarray(As[:, sd - small.red[sd]].todense()).ravel()
        for di in (-1, 0, 1):
            for dj in (-1, 0, 1):
                for dk in (-1, 0, 1):
                    for d2 in range(3):
                        gb = job.node_dof.reshape(-1, 3)[node(n, 200 + di, 200 + dj, 200 + dk), d2]
                        gs = small.node_dof.reshape(-1, 3)[node(6, 3 + di, 3 + dj, 3 + dk), d2]
                        assert abs(col[gb - job.red[gb]] - ocol[gs - small.red[gs]]) <= 1e-12 * np.abs(ocol).max()
        assert abs(np.abs(col).sum() - np.abs(ocol).sum()) <= 1e-11 * np.abs(ocol).sum()    # nothing outside the stencil
    K.free()
    t_g1 = time.time() - t0
    # ---- G2, mixed precision, to 1e-8, independent residual
    job2 = problem.cube_job(n)
    gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
    try:
        K = gpu_ctx.assemble_hex8(job2.xyz, job2.node_dof, job2.conn, job2.elem_mat, job2.elem_type, job2.mat_E_nu, job2.red)
        gpu_ctx.set_profiling(True)
        U, rep = K.cg_solve(job2.F, 1e-8, precision_mode=hip.PREC_MIXED)
    finally:
        gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 1)
        gpu_ctx.set_profiling(False)
    prof = gpu_ctx.profile()
    # Round 5: the library says what it delivered.  The loop iterates on the fp32 copy; rel_residual is the residual of
    # the returned point under the FP64 matrix (one extra product), and the default (STAN_OPT_CG_REFINE = 1) keeps
    # refining until THAT meets eps.  Checked against an independent figure: the library's plain product on the unscaled
    # matrix and the exported diagonal, combined in numpy (Matrix.scaled_residual).
    assert rep["terminationtype"] == 1 and rep["rel_residual"] <= 1e-8, rep
    r_true = K.scaled_residual(job2.F, U)
    print("400^3: mesh %.0f s, G1 part %.0f s, total %.0f s; mixed solve %d its in %d pass(es), recurrence residual %.2e, "
          "fp64 residual reported %.3e, independent %.3e" %
          (t_mesh, t_g1, time.time() - t0, rep["iterations"], prof["refine_passes"], prof["rel_residual_recurrence"],
           rep["rel_residual"], r_true))
    assert abs(r_true - rep["rel_residual"]) <= 0.1 * rep["rel_residual"] + 1e-12   # (two fp64 products in different orders)
    assert r_true <= 1.1e-8
    assert prof["rel_residual_fp64"] == rep["rel_residual"] and prof["fp64_products"] >= 1
    assert 2000 <= rep["iterations"] <= 12000
    K.free()
