"""GPU parity tests: libstan_hip.so (through the C-ABI) against the CPU oracle on the same
seeded inputs, plus size-independent properties at larger sizes.

Bars: DOF / CSR indexing bit-exact; K values <= 1e-13 relative to max|K| (fp64, different
summation order than MatrixST); displacements <= 1e-6 relative (BASELINE.json north_star)."""
import os

import numpy as np
import pytest

from stan_amd import problem
from tests.util import UNIT, random_hexes

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "hot_path_golden.npz"))
K_TOL = 1e-13
U_TOL = 1e-6


def _assemble_both(ctx, oracle, job):
    K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                          job.mat_E_nu, job.red)
    rc, A = oracle.assemble(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                            job.mat_E_nu, job.red)
    assert rc == 0
    return K, A


@pytest.mark.parametrize("etype", [1, 2])
def test_ke_parity(gpu_ctx, oracle, etype):
    xs = random_hexes(64, seed=21)
    Kg = gpu_ctx.ke_hex8_batch(xs, 70000.0, 0.33, np.full(64, etype, np.uint8))
    for x, kg in zip(xs, Kg):
        rc, ko = oracle.ke_hex8(x, 70000.0, 0.33, etype)
        assert rc == 0
        assert np.abs(kg - ko).max() <= K_TOL * np.abs(ko).max()
    k1 = gpu_ctx.ke_hex8(UNIT, 210000.0, 0.3, etype)
    assert np.abs(k1 - GOLD["ke_unit_g%d" % etype]).max() <= K_TOL * np.abs(k1).max()


def test_ke_singular_jacobian_is_an_error(gpu_ctx):
    from stan_amd import hip
    x = UNIT.copy(); x[:, 2] = 0.0
    with pytest.raises(hip.StanHipError) as ei:
        gpu_ctx.ke_hex8(x, 1.0, 0.3, 2)
    assert ei.value.code == hip.E_DETJ and gpu_ctx.last_bad_element() == 0


@pytest.mark.parametrize("n,etype,jit", [(1, 2, 0.0), (2, 2, 0.0), (3, 2, 0.1), (5, 1, 0.1),
                                         (8, 2, 0.1), (13, 2, 0.05)])
def test_assembly_csr_parity(gpu_ctx, oracle, n, etype, jit):
    job = problem.cube_job(n, etype=etype, jitter=jit)
    K, A = _assemble_both(gpu_ctx, oracle, job)
    info = K.info()
    assert info["n_dof"] == job.n_dof and info["n_reduced"] == job.n_red
    rowptr, col, val = K.to_csr(upper_only=True)
    assert np.array_equal(rowptr, A.ridx)           # bit-exact indexing
    assert np.array_equal(col, A.idx)
    assert np.abs(val - A.vals).max() <= K_TOL * np.abs(A.vals).max()
    # full (symmetric) export is the mirror of the upper one
    rp, cf, vf = K.to_csr(upper_only=False)
    assert rp[-1] == 2 * A.nnz - A.n
    K.free()


def test_assembly_is_bit_reproducible(gpu_ctx):
    job = problem.cube_job(6, jitter=0.1)
    vals = []
    for _ in range(2):
        K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                                  job.mat_E_nu, job.red)
        vals.append(K.to_csr()[2])
        K.free()
    assert np.array_equal(vals[0], vals[1])


def test_assembly_mixed_types_materials_partial_spc(gpu_ctx, oracle):
    rng = np.random.default_rng(3)
    job = problem.cube_job(5, jitter=0.1)
    job.elem_type = rng.integers(1, 3, job.conn.shape[0]).astype(np.uint8)
    job.elem_mat = rng.integers(0, 3, job.conn.shape[0]).astype(np.int32)
    job.mat_E_nu = np.array([[210000.0, 0.3], [70000.0, 0.33], [1000.0, 0.45]])
    # partially fixed nodes: (1,0,1) on a few nodes
    from stan_amd import host
    from stan_amd.cube import cube_bcs
    spc, ld, f = cube_bcs(5)
    vals = np.ones((spc.shape[0], 3)); vals[::3, 1] = 0
    job.red, job.n_fixed = host.dof_reduction(job.n_dof, job.node_dof, spc, vals)
    K, A = _assemble_both(gpu_ctx, oracle, job)
    rowptr, col, val = K.to_csr()
    assert np.array_equal(rowptr, A.ridx) and np.array_equal(col, A.idx)
    assert np.abs(val - A.vals).max() <= K_TOL * np.abs(A.vals).max()
    K.free()


def test_assembly_shuffled_wire_order(gpu_ctx, oracle):
    """Node and element wire order is arbitrary in an STdb (Dictionary order)."""
    from stan_amd.cube import cube_bcs, cube_mesh
    rng = np.random.default_rng(9)
    n = 4
    xyz, conn = cube_mesh(n, jitter=0.1)
    perm = rng.permutation(xyz.shape[0])
    xyz2 = np.empty_like(xyz); xyz2[perm] = xyz
    conn2 = perm[conn][rng.permutation(conn.shape[0])].astype(np.int32)
    spc, ld, f = cube_bcs(n)
    job = problem.make_job(xyz2, conn2, perm[spc], np.ones((spc.shape[0], 3)), perm[ld],
                           np.tile(f, (ld.shape[0], 1)))
    K, A = _assemble_both(gpu_ctx, oracle, job)
    rowptr, col, val = K.to_csr()
    assert np.array_equal(rowptr, A.ridx) and np.array_equal(col, A.idx)
    assert np.abs(val - A.vals).max() <= K_TOL * np.abs(A.vals).max()
    U, rep = K.cg_solve(job.F, 1e-12)
    Uo, _ = oracle.cg(A, job.F, 1e-12)
    assert np.abs(U - Uo).max() <= U_TOL * np.abs(Uo).max()
    K.free()


def test_assembly_errors(gpu_ctx):
    from stan_amd import hip
    job = problem.cube_job(3)
    xyz = job.xyz.copy()
    xyz[job.conn[5]] = xyz[job.conn[5]] * [1, 1, 0]       # flatten one element: det J == 0
    with pytest.raises(hip.StanHipError) as ei:
        gpu_ctx.assemble_hex8(xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                              job.mat_E_nu, job.red)
    assert ei.value.code == hip.E_DETJ
    bad_dof = job.node_dof.copy(); bad_dof[4, 1] += 7
    with pytest.raises(hip.StanHipError) as ei:
        gpu_ctx.assemble_hex8(job.xyz, bad_dof, job.conn, job.elem_mat, job.elem_type,
                              job.mat_E_nu, job.red)
    assert ei.value.code == hip.E_DOF_LAYOUT
    bad_t = job.elem_type.copy(); bad_t[0] = 9
    with pytest.raises(hip.StanHipError) as ei:
        gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, bad_t,
                              job.mat_E_nu, job.red)
    assert ei.value.code == hip.E_UNSUPPORTED


def test_spmv_parity(gpu_ctx, oracle):
    job = problem.cube_job(7, jitter=0.1)
    K, A = _assemble_both(gpu_ctx, oracle, job)
    x = np.random.default_rng(0).standard_normal(job.n_red)
    y = K.spmv(x)
    yo = oracle.smv_upper(A, x)
    assert np.abs(y - yo).max() <= 1e-12 * np.abs(yo).max()
    K.free()


@pytest.mark.parametrize("tag,n,etype,jit", [("cube2_g2_j0", 2, 2, 0.0), ("cube4_g2_j1", 4, 2, 0.1),
                                              ("cube6_g2_j0", 6, 2, 0.0), ("cube5_g1_j1", 5, 1, 0.1)])
def test_cg_vs_golden_direct_solve(gpu_ctx, tag, n, etype, jit):
    job = problem.cube_job(n, etype=etype, jitter=jit)
    K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                              job.mat_E_nu, job.red)
    U, rep = K.cg_solve(job.F, 1e-12)
    assert rep["terminationtype"] in (1, 7)
    Ud = GOLD[tag + "_U"]
    assert np.abs(U - Ud).max() <= U_TOL * np.abs(Ud).max()
    K.free()


@pytest.mark.parametrize("n,etype", [(10, 2), (16, 2), (12, 1)])
def test_cg_parity_with_oracle(gpu_ctx, oracle, n, etype):
    job = problem.cube_job(n, etype=etype, jitter=0.05)
    K, A = _assemble_both(gpu_ctx, oracle, job)
    for eps in (1e-12, 1e-8):
        U, rep = K.cg_solve(job.F, eps)
        Uo, repo = oracle.cg(A, job.F, eps)
        assert rep["terminationtype"] == repo["terminationtype"]
        # Both sides stop within kappa * eps of the solution of the same system, so they are within
        # 2 kappa eps of each other.  kappa = condition number of the Jacobi-scaled matrix S K S, measured:
        # 12.7 n^2 for the x-clamped G2 cube (tools/cpu_sizes.py, profiles/r02/cpu_sizes_n100_*.jsonl),
        # 2.4 n^4 for G1, which has no hourglass control (profiles/r02/CONFIG5.md).  (Rounds 1-2 allowed
        # 1e-3 / 2e-2 here; the same recurrences in fact agree far better: 3e-11 at 56^3, 8.5e-11 at 148^3.)
        # A solve that ends on alglib's merit-function floor (type 7) has not reached eps: its distance from the
        # solution is kappa times the residual it DID reach (both sides report it).
        kappa = 12.7 * n * n if etype == 2 else 2.4 * n ** 4
        tol = max(U_TOL, kappa * (max(eps, rep["rel_residual"]) + max(eps, repo["rel_residual"])))
        assert np.abs(U - Uo).max() <= tol * np.abs(Uo).max(), (np.abs(U - Uo).max() / np.abs(Uo).max(), tol)
        # same algorithm => iteration counts agree up to rounding-induced drift
        # (a type-7 stop sits on the rounding floor, where the count is noise-dependent)
        slack = 10 if rep["terminationtype"] == 1 else 4
        assert abs(rep["iterations"] - repo["iterations"]) <= max(5, repo["iterations"] // slack)
    K.free()


def test_cg_termination_codes(gpu_ctx, oracle):
    job = problem.cube_job(4)
    K, A = _assemble_both(gpu_ctx, oracle, job)
    U, rep = K.cg_solve(job.F, 1e-30, max_its=5)
    Uo, repo = oracle.cg(A, job.F, 1e-30, maxits=5)
    assert rep["terminationtype"] == 5 and rep["iterations"] == 5
    assert np.abs(U - Uo).max() <= 1e-9 * np.abs(Uo).max()
    U, rep = K.cg_solve(job.F, 0.0, max_its=0)       # both zero -> eps_f = 1e-6
    assert rep["terminationtype"] == 1 and rep["rel_residual"] <= 1e-6
    U, rep = K.cg_solve(job.F, 1e-30)                 # unreachable: type 7, best point
    assert rep["terminationtype"] == 7
    Uo, _ = oracle.cg(A, job.F, 1e-12)
    assert np.abs(U - Uo).max() <= U_TOL * np.abs(Uo).max()
    U, rep = K.cg_solve(np.zeros_like(job.F), 1e-8)
    assert rep["terminationtype"] == 1 and rep["iterations"] == 0 and not U.any()
    K.free()


def test_cg_type7_on_a_slender_model_like_the_reference_screenshot(gpu_ctx, oracle):
    """images/Solver.PNG: the reference's own 43 650-DOF run ends with type 7 at tolerance 1e-6.
    A slender 115 x 10 x 10 cantilever (42 108 DOF) does the same here and in the oracle: the
    merit-function rule fires before the residual test, and U is returned regardless."""
    nx, ny, nz = 115, 10, 10
    mx, my, mz = nx + 1, ny + 1, nz + 1
    k, j, i = np.meshgrid(np.arange(mz), np.arange(my), np.arange(mx), indexing="ij")
    xyz = np.stack([i.ravel(), j.ravel(), k.ravel()], axis=1).astype(np.float64)
    ke, je, ie = (v.ravel() for v in np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij"))
    nid = lambda a, b, c: a + mx * (b + my * c)  # noqa: E731
    conn = np.stack([nid(ie, je, ke), nid(ie + 1, je, ke), nid(ie + 1, je + 1, ke), nid(ie, je + 1, ke),
                     nid(ie, je, ke + 1), nid(ie + 1, je, ke + 1), nid(ie + 1, je + 1, ke + 1),
                     nid(ie, je + 1, ke + 1)], axis=1).astype(np.int32)
    spc = np.nonzero(xyz[:, 0] == 0)[0].astype(np.int32)
    ld = np.nonzero(xyz[:, 0] == nx)[0].astype(np.int32)
    job = problem.make_job(xyz, conn, spc, np.ones((len(spc), 3)), ld, np.tile([0.0, 0.0, 50.0], (len(ld), 1)))
    K, A = _assemble_both(gpu_ctx, oracle, job)
    U, rep = K.cg_solve(job.F, 1e-6)
    Uo, repo = oracle.cg(A, job.F, 1e-6)
    assert rep["terminationtype"] == repo["terminationtype"] == 7
    assert rep["rel_residual"] > 1e-6 and np.isfinite(U).all() and np.abs(U).max() > 0
    # where exactly the merit function first ticks up is decided by rounding (different summation
    # orders on the two sides): same order of iterations, both short of the tolerance
    assert 0.5 <= rep["iterations"] / repo["iterations"] <= 2.0
    K.free()


def test_cg_not_spd_is_reported_not_raised(gpu_ctx):
    # E < 0 makes K negative definite: ALGLIB reports -5 and STAN still returns U
    job = problem.cube_job(3, E=-210000.0)
    K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                              job.mat_E_nu, job.red)
    U, rep = K.cg_solve(job.F, 1e-8)
    assert rep["terminationtype"] == -5
    K.free()


def test_cg_mixed_precision(gpu_ctx, oracle):
    from stan_amd import hip
    job = problem.cube_job(10, jitter=0.05)
    K, A = _assemble_both(gpu_ctx, oracle, job)
    U, rep = K.cg_solve(job.F, 1e-8, precision_mode=hip.PREC_MIXED)
    Uo, _ = oracle.cg(A, job.F, 1e-12)
    # fp32 matrix entries: solution of a 6e-8-perturbed system
    assert rep["terminationtype"] in (1, 7)
    assert np.abs(U - Uo).max() <= 1e-4 * np.abs(Uo).max()
    K.free()


def test_cg_fixed48_stream(gpu_ctx, oracle):
    """STAN_PREC_FIXED48: fp64 arithmetic on the 48-bit fixed-point copy of S K S (|a| <= 1).
    Entries move by <= 2^-47 of the unit diagonal, so U stays inside the north-star tolerance
    with a wide margin and the iteration history is that of the fp64 stream."""
    from stan_amd import hip
    job = problem.cube_job(12, jitter=0.1)
    job.elem_mat = (np.arange(job.conn.shape[0]) % 2).astype(np.int32)
    job.mat_E_nu = np.array([[210000.0, 0.3], [7000.0, 0.33]])   # kappa(S K S) ~ 1.2e4
    gpu_ctx.set_profiling(True)
    gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)   # converge both runs to the residual test
    try:
        K, A = _assemble_both(gpu_ctx, oracle, job)
        vals_fresh = K.to_csr()[2]          # export BEFORE any solve: the assembled bits, and K stays as assembled
        assert K.info()["scaled"] == 0 and np.array_equal(K.to_csr()[2], vals_fresh)
        # at a tolerance the quantised entries can carry (7e-15 * kappa ~ 1e-10 in the residual) the iteration history is
        # the fp64 stream's and the fp64 check passes at once
        U64c, rep64c = K.cg_solve(job.F, 1e-9, 20000)
        U48c, rep48c = K.cg_solve(job.F, 1e-9, 20000, precision_mode=hip.PREC_FIXED48)
        assert gpu_ctx.profile()["refine_passes"] == 1 and rep48c["rel_residual"] <= 1e-9
        assert rep48c["terminationtype"] == rep64c["terminationtype"] == 1
        assert abs(rep48c["iterations"] - rep64c["iterations"]) <= max(2, rep64c["iterations"] // 50)
        # below that floor the recurrence alone would claim 1e-12 for a point whose fp64 residual is ~1e-10: since
        # round 5 the solve checks, and a refinement pass (STAN_OPT_CG_REFINE) delivers what was asked for
        U64, rep64 = K.cg_solve(job.F, 1e-12, 20000)
        U48, rep48 = K.cg_solve(job.F, 1e-12, 20000, precision_mode=hip.PREC_FIXED48)
        assert gpu_ctx.profile()["value_stream"] == hip.PREC_FIXED48
        assert gpu_ctx.profile()["spmv_bytes"] < 0.8 * 76 * K.info()["n_blocks"] + 20 * job.n_red
        assert gpu_ctx.profile()["refine_passes"] >= 2 and rep48["rel_residual"] <= 1e-12
    finally:
        gpu_ctx.set_profiling(False)
        gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 1)
    assert rep48["terminationtype"] == rep64["terminationtype"] == 1
    assert rep64["iterations"] <= rep48["iterations"] <= 2 * rep64["iterations"]
    # quantisation alone moves U by 4e-12 here (direct solves of both matrices on the CPU);
    # the bound leaves room for the two CG runs' own kappa * 1e-12
    assert np.abs(U48 - U64).max() <= 1e-8 * np.abs(U64).max()
    Uo, _ = oracle.cg(A, job.F, 1e-12)
    assert np.abs(U48 - Uo).max() <= 1e-6 * np.abs(Uo).max()
    assert K.spmv_bench(3, hip.PREC_FIXED48) > 0
    # the fp64 values are untouched by four solves and by the exports: an export after them divides the scaled values on
    # its way out ((a t) / t: within 1 ulp of the export before any solve, which is the assembled bits -- round 6), twice
    # the same bits, and the oracle's matrix
    rowptr, cols, vals = K.to_csr()
    assert np.abs(vals - vals_fresh).max() <= 2.3e-16 * np.abs(vals_fresh).max() and np.all(np.abs(vals - vals_fresh) <= 2.3e-16 * np.abs(vals_fresh))
    assert np.array_equal(K.to_csr()[2], vals)
    assert np.abs(vals - A.vals).max() <= K_TOL * np.abs(A.vals).max()
    K.free()


def test_cg_fixed48_falls_back_when_not_spd(gpu_ctx):
    """E < 0 makes K negative definite: the scaling leaves entries far outside [-1, 1], the
    fixed-point copy is refused and the fp64 stream reports ALGLIB's -5 as before."""
    from stan_amd import hip
    job = problem.cube_job(4)
    job.mat_E_nu = np.array([[-210000.0, 0.3]])
    K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                              job.mat_E_nu, job.red)
    gpu_ctx.set_profiling(True)
    try:
        U, rep = K.cg_solve(job.F, 1e-8, precision_mode=hip.PREC_FIXED48)
        assert gpu_ctx.profile()["value_stream"] == hip.PREC_FP64
    finally:
        gpu_ctx.set_profiling(False)
    assert rep["terminationtype"] == -5
    K.free()


def test_block_pool_reuses_and_releases_device_memory(built_libs):
    """STAN_OPT_POOL: blocks >= 8 MB freed by the library stay with the context and are handed out
    again (a hipMalloc of tens of GB costs 0.4-1.8 s on this stack); switching the option off gives
    the memory back; results do not depend on it."""
    import torch  # noqa: F401
    from stan_amd import hip
    job = problem.cube_job(40)
    args = (job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    ctx = hip.Context(0)
    assert ctx.pool_info() == (0, 0)
    res, parked = [], []
    for i in range(3):
        K = ctx.assemble_hex8(*args)
        res.append(K.cg_solve(job.F, 1e-8))
        assert ctx.pool_info()[0] < 64e6 or i > 0       # while K lives, its big blocks are out of the pool
        K.free()
        parked.append(ctx.pool_info())
    assert parked[0][0] > 130e6                        # the 134 MB value array and friends are parked
    assert parked[1] == parked[0] and parked[2] == parked[0]   # cycles 2 and 3 allocated nothing new
    assert all(np.array_equal(res[0][0], r[0]) and res[0][1] == r[1] for r in res[1:])
    ctx.set_option(hip.OPT_POOL, 0)                    # flush + plain hipMalloc/hipFree from here on
    assert ctx.pool_info() == (0, 0)
    K = ctx.assemble_hex8(*args)
    U, rep = K.cg_solve(job.F, 1e-8)
    K.free()
    assert np.array_equal(U, res[0][0]) and ctx.pool_info() == (0, 0)
    ctx.close()


def test_allocation_by_trial_does_not_change_results(built_libs, oracle):
    """STAN_OPT_PLACEMENT_TRIES > 1 (placement.hip): K's value array is the fastest-streaming of
    several hipMalloc blocks; the matrix and the solution are the same bits as with plain allocation."""
    import torch  # noqa: F401
    from stan_amd import hip
    job = problem.cube_job(72, jitter=0.05)       # value array 0.74 GB: above the 256 MB threshold
    args = (job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    out = []
    for tries in (1, 3):
        ctx = hip.Context(0)
        ctx.set_option(hip.OPT_PLACEMENT_TRIES, tries)
        K = ctx.assemble_hex8(*args)
        y = K.spmv_local(np.arange(job.n_dof, dtype=np.float64) % 7 - 3.0)   # before the CG scales K
        U, rep = K.cg_solve(job.F, 1e-8)
        out.append((U, rep, y))
        K.free()
        K = ctx.assemble_hex8(*args)              # second assembly: the pooled winner, no new search
        assert np.array_equal(K.spmv_local(np.arange(job.n_dof, dtype=np.float64) % 7 - 3.0), y)
        K.free()
        ctx.close()
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1] and np.array_equal(out[0][2], out[1][2])
    with pytest.raises(hip.StanHipError):
        hip.Context(0).set_option(hip.OPT_PLACEMENT_TRIES, 99)


def test_matrix_may_outlive_its_context(built_libs):
    """Destroying a context detaches its matrices: freeing them afterwards is safe (their buffers
    go straight back to the driver)."""
    import torch  # noqa: F401
    from stan_amd import hip
    job = problem.cube_job(6)
    ctx = hip.Context(0)
    K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    K2 = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    K2.free()
    lib, h = ctx.lib, ctx.h
    lib.stan_hip_destroy(h)      # the context goes first
    ctx.h = None
    lib.stan_hip_matrix_free(K.k)
    K.k = None


def test_cg_is_bit_reproducible(gpu_ctx):
    job = problem.cube_job(8, jitter=0.1)
    res = []
    for _ in range(2):
        K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                                  job.mat_E_nu, job.red)
        res.append(K.cg_solve(job.F, 1e-10))
        K.free()
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]


def test_medium_cube_properties(gpu_ctx):
    """40^3 (~207k DOF): too slow to cross-check entry by entry; size-independent properties."""
    n = 40
    job = problem.cube_job(n)
    K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                              job.mat_E_nu, job.red)
    info = K.info()
    # SURVEY.md section 8 formula on the UNREDUCED block matrix: (3n+1)^3-ish blocks
    assert info["n_block_rows"] == (n + 1) ** 3
    assert info["n_blocks"] == (3 * n + 1) ** 3
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(job.n_red), rng.standard_normal(job.n_red)
    Kx, Ky = K.spmv(x), K.spmv(y)
    assert abs(y @ Kx - x @ Ky) <= 1e-10 * abs(y @ Kx)              # symmetry
    assert np.abs(K.spmv(2 * x - 3 * y) - (2 * Kx - 3 * Ky)).max() <= 1e-10 * np.abs(Kx).max()
    assert x @ Kx > 0                                                # positive definite
    from stan_amd import hip
    gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)    # ALGLIB's type-7 floor sits near 1e-7 here
    try:
        U, rep = K.cg_solve(job.F, 1e-10)
    finally:
        gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 1)
    assert rep["terminationtype"] == 1 and rep["rel_residual"] <= 1e-10
    r = job.F - K.spmv(U)                                            # independent residual
    assert np.linalg.norm(r) <= 1e-7 * np.linalg.norm(job.F)
    # total reaction balances the applied load: sum of F_z = 50 * (n+1)^2
    assert np.isclose(job.F.sum(), 50.0 * (n + 1) ** 2)
    K.free()


def test_stress_recovery_parity(gpu_ctx, oracle):
    job = problem.cube_job(5, jitter=0.1)
    job.elem_mat = (np.arange(job.conn.shape[0]) % 2).astype(np.int32)
    job.mat_E_nu = np.array([[210000.0, 0.3], [70000.0, 0.33]])
    rng = np.random.default_rng(4)
    disp = rng.standard_normal(job.xyz.shape) * 1e-3
    strain, stress = gpu_ctx.recover_hex8(job.xyz, disp, job.conn, job.elem_mat, job.elem_type,
                                          job.mat_E_nu)
    for e in range(job.conn.shape[0]):
        E, nu = job.mat_E_nu[job.elem_mat[e]]
        rc, eo, so = oracle.recover_hex8(job.xyz[job.conn[e]], E, nu, 2, disp[job.conn[e]].ravel())
        assert rc == 0
        assert np.abs(strain[e] - eo).max() <= 1e-12 * np.abs(eo).max()
        assert np.abs(stress[e] - so).max() <= 1e-12 * np.abs(so).max()


def test_recovery_results_kept_on_the_device_and_mapped_by_threads(gpu_ctx):
    """stan_hip_recover_hex8_keep + stan_hip_results_map (what the console driver's export uses): any element range, from
    several host threads at once, gives the rows stan_hip_recover_hex8 returns -- bit for bit; 7^3 = 343 elements is not a
    multiple of the 8 elements a wavefront handles (the kernel's ragged tail)."""
    import threading
    job = problem.cube_job(7, jitter=0.1)
    disp = np.random.default_rng(11).standard_normal(job.xyz.shape) * 1e-3
    strain, stress = gpu_ctx.recover_hex8(job.xyz, disp, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu)
    res = gpu_ctx.recover_hex8_keep(job.xyz, disp, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu)
    ne = job.conn.shape[0]
    assert res.n_elem == ne == 343
    for a, b in ((0, ne), (0, 0), (5, 17), (ne - 3, ne), (100, 101)):
        e, s = res.map(a, b)
        assert np.array_equal(e, strain[a:b]) and np.array_equal(s, stress[a:b]), (a, b)
    bad = []

    def worker(t):
        rng = np.random.default_rng(t)
        for _ in range(50):
            a = int(rng.integers(0, ne))
            b = int(rng.integers(a, ne + 1))
            e, s = res.map(a, b)
            if not (np.array_equal(e, strain[a:b]) and np.array_equal(s, stress[a:b])):
                bad.append((t, a, b))
    th = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not bad, bad[:3]
    from stan_amd import hip
    with pytest.raises(hip.StanHipError):
        res.map(0, ne + 1)
    res.free()


def test_stress_recovery_at_100_cubed_on_sampled_elements(gpu_ctx, oracle):
    """k_recover at a size of BASELINE.json (config 2: 10^6 elements; 125 000 wavefronts of 8 elements) against the oracle's
    Element.Recovery_Stress (Element.cs:211-246) on 400 sampled elements -- corners, faces, the last ones of the ragged tail
    and random interior ones -- with the displacements of a real solve; both entry points (host arrays / kept on the device)."""
    n = 100
    job = problem.cube_job(n)
    gpu_ctx.set_option(1, 0)
    try:
        K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
        U, rep = K.cg_solve(job.F, 1e-8)
        K.free()
    finally:
        gpu_ctx.set_option(1, 1)
    assert rep["terminationtype"] == 1
    disp_full = np.zeros(job.n_dof)
    disp_full[job.red != -1] = U
    disp = disp_full[job.node_dof]                     # [n_nodes, 3] in NodeLib order
    strain, stress = gpu_ctx.recover_hex8(job.xyz, disp, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu)
    res = gpu_ctx.recover_hex8_keep(job.xyz, disp, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu)
    ne = job.conn.shape[0]
    rng = np.random.default_rng(100)
    sample = np.unique(np.concatenate([[0, 1, 7, 8, n - 1, n * n - 1, ne - 9, ne - 8, ne - 2, ne - 1], rng.integers(0, ne, 390)]))
    E, nu = job.mat_E_nu[0]
    smax, emax = np.abs(stress).max(), np.abs(strain).max()
    for e in sample:
        rc, eo, so = oracle.recover_hex8(job.xyz[job.conn[e]], E, nu, 2, disp[job.conn[e]].ravel())
        assert rc == 0
        assert np.abs(strain[e] - eo).max() <= 1e-12 * emax, e     # (differences of displacements of size 0.26: absolute bars)
        assert np.abs(stress[e] - so).max() <= 1e-12 * smax, e
    for a, b in ((0, 4096), (ne - 4097, ne), (123457, 131072)):
        e2, s2 = res.map(a, b)
        assert np.array_equal(e2, strain[a:b]) and np.array_equal(s2, stress[a:b])
    res.free()


def test_nodal_forces_parity(gpu_ctx, oracle):
    """Element.Compute_NodalForces + the R assembly of Solver.cs:184-196 against the oracle."""
    job = problem.cube_job(5, jitter=0.1)
    job.elem_mat = (np.arange(job.conn.shape[0]) % 2).astype(np.int32)
    job.mat_E_nu = np.array([[210000.0, 0.3], [70000.0, 0.33]])
    disp = np.random.default_rng(8).standard_normal(job.xyz.shape) * 1e-3
    f, R = gpu_ctx.nodal_forces_hex8(job.xyz, disp, job.node_dof, job.conn, job.elem_mat,
                                     job.elem_type, job.mat_E_nu)
    fo = np.zeros_like(f)
    for e in range(job.conn.shape[0]):
        E, nu = job.mat_E_nu[job.elem_mat[e]]
        rc, _, so = oracle.recover_hex8(job.xyz[job.conn[e]], E, nu, 2, disp[job.conn[e]].ravel())
        rc2, fo[e] = oracle.nodal_forces_hex8(job.xyz[job.conn[e]], 2, so)
        assert rc == 0 and rc2 == 0
    assert np.abs(f - fo).max() <= 1e-12 * np.abs(fo).max()
    Ro = np.zeros(job.n_dof)
    np.add.at(Ro, job.node_dof[job.conn].reshape(-1, 24), fo)
    assert np.abs(R - Ro).max() <= 1e-12 * np.abs(Ro).max()


def test_stress_recovery_g1_is_an_error_like_the_reference(gpu_ctx):
    from stan_amd import hip
    job = problem.cube_job(2, etype=1)
    with pytest.raises(hip.StanHipError) as ei:
        gpu_ctx.recover_hex8(job.xyz, np.zeros_like(job.xyz), job.conn, job.elem_mat, job.elem_type,
                             job.mat_E_nu)
    assert ei.value.code == hip.E_UNSUPPORTED and gpu_ctx.last_bad_element() == 0


def test_native_console_driver_end_to_end(built_libs, oracle, tmp_path):
    """stan_solver <model.STdb>: the Solver.Main replacement over both C-ABI libraries."""
    import subprocess
    from stan_amd import bdf, host
    from stan_amd.cube import cube_bcs, cube_mesh
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "stan_amd", "bin", "stan_solver")
    n = 6
    xyz, conn = cube_mesh(n, jitter=0.1)
    bdf.write_bdf(str(tmp_path / "m.bdf"), xyz, conn)
    d = host.Db()
    assert d.read_bdf(str(tmp_path / "m.bdf")) == 0
    d.add_material(1, "Steel", 210000.0, 0.3)
    d.assign_part(1, 1, "HEX8_G2")
    spc, ld, f = cube_bcs(n)
    d.add_bc(1, "fix", "SPC", spc + 1, np.ones((len(spc), 3)))
    d.add_bc(2, "load", "PointLoad", ld + 1, np.tile(f, (len(ld), 1)))
    d.set_analysis(tol=1e-12)
    path = str(tmp_path / "model.STdb")
    d.write_stdb(path)
    out = subprocess.run([exe, "--json", path], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "NORMAL" in out.stdout and "Stress recovery" in out.stdout
    import json
    summary = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert summary["n_dof"] == 3 * (n + 1) ** 3 and summary["termination_type"] in (1, 7)
    assert summary["cg_iterations"] > 0 and summary["spmv_ms"] > 0
    r = host.Db.read_stdb(path)
    assert r.sizes()["result_step"] == 1
    disp, strain, stress = r.results(1)
    # oracle on the model as the driver saw it (coordinates as parsed from the bdf)
    m = host.Db.read_stdb(path); m.assign_dof()
    fl = m.flat(); red, nfix, F = m.reduction()
    rc, A = oracle.assemble(fl["xyz"], fl["node_dof"], fl["conn"], fl["elem_mat"], fl["elem_type"],
                            fl["mat_E_nu"], red)
    Uo, _ = oracle.cg(A, F, 1e-12)
    do = host.nodal_displacements(fl["node_dof"], red, Uo)
    assert np.abs(disp - do).max() <= U_TOL * np.abs(do).max()
    for e in (0, 17, n ** 3 - 1):
        rc, eo, so = oracle.recover_hex8(fl["xyz"][fl["conn"][e]], 210000.0, 0.3, 2, do[fl["conn"][e]].ravel())
        assert np.abs(stress[e] - so).max() <= 1e-5 * np.abs(so).max()
    # a G1 model fails in stress recovery like the reference and leaves the input untouched
    d.assign_part(1, 1, "HEX8_G1")
    d.write_stdb(path)
    before = open(path, "rb").read()
    out = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "HEX8_G1" in out.stderr
    assert open(path, "rb").read() == before


def test_plain_c_consumer_end_to_end(built_libs, oracle, tmp_path):
    """A C99 program written against include/*.h only (tests/c_abi/consumer.c) runs AssignDOF ->
    reduction -> assemble -> CG -> write-back -> stress recovery; its displacements equal the
    oracle's."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "c_abi", "consumer")
    n = 6
    out = subprocess.run([exe, str(n), str(tmp_path / "disp.bin")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "NORMAL" in out.stdout
    disp = np.fromfile(str(tmp_path / "disp.bin"), dtype=np.float64).reshape(-1, 3)
    job = problem.cube_job(n)
    rc, A = oracle.assemble(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    Uo, rep = oracle.cg(A, job.F, 1e-10)
    full = oracle.include_bc(job.red, Uo)
    assert np.abs(disp - full[job.node_dof]).max() <= 1e-6 * np.abs(Uo).max()


@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_shards_and_halo_plan_on_one_gpu(built_libs, nranks):
    """Every rank's shard, assembled as a DETACHED rank on this GPU: the device-derived halo
    plan equals the host plan (tests/test_distributed.py runs the sharded CG on it over gloo)
    and shard x [owned | halo] reproduces the rows of the unsharded product bit for bit."""
    import torch  # noqa: F401
    from stan_amd import hip, host
    job = problem.cube_job(11, jitter=0.1)
    args = (job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    ctx = hip.Context(0)
    K1 = ctx.assemble_hex8(*args)
    x = np.random.default_rng(2).standard_normal(job.n_dof)
    y_ref = K1.spmv_local(x)
    K1.free()
    for r in range(nranks):
        ctx.comm_init(r, nranks, None)
        K = ctx.assemble_hex8(*args)
        dev, ref = K.plan(), host.partition_plan(job.node_index, job.conn, nranks, r)
        assert np.array_equal(dev["row_starts"][:nranks + 1], ref["row_starts"])
        for k in ("halo_glob", "nbr", "send_off", "recv_off", "send_rows"):
            assert np.array_equal(dev[k], ref[k]), k
        r0, r1 = dev["row_begin"], dev["row_end"]
        xb = x.reshape(-1, 3)
        x_local = np.concatenate([xb[r0:r1], xb[dev["halo_glob"]]]).ravel()
        assert np.array_equal(K.spmv_local(x_local), y_ref[3 * r0:3 * r1])
        with pytest.raises(hip.StanHipError) as ei:      # a detached rank cannot run collectives
            K.cg_solve(job.F, 1e-8)
        assert ei.value.code == hip.E_COMM
        K.free()
    ctx.close()


def test_single_rank_rccl_communicator(built_libs):
    """The RCCL code path (dlopen, all-reduce, broadcast-gather) on a real 1-rank communicator
    gives the same bits as the communicator-free path (run in a child: tests/rccl_single_worker.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_single_worker.py")],
                         capture_output=True, text=True, timeout=300, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("same_bits")][0].split()
    assert line[1] == "1" and line[3] == line[4] and int(line[3]) > 0
    # round 6: ONE RCCL per process.  torch has mapped its bundled librccl.so before comm_init asks; the library
    # takes that file (RTLD_NOLOAD) instead of loading /opt/rocm's next to it, and says which file it is.
    rl = [l for l in out.stdout.splitlines() if l.startswith("rccl ")][0].split()
    assert "librccl" in rl[1] and os.path.isabs(rl[1]) and rl[3] == "1" and rl[5] == "1" and int(rl[7]) > 20000, out.stdout
    mapped = [l for l in out.stdout.splitlines() if l.startswith("mapped ")][0].split()[1:]
    assert [os.path.realpath(m) for m in mapped] == [rl[1]]


def _star_job(k, layers=3, rings=2):
    from stan_amd.cube import star_mesh
    xyz, conn = star_mesh(k, layers, rings)
    z0 = np.nonzero(xyz[:, 2] == 0)[0]
    top = np.nonzero(xyz[:, 2] == xyz[:, 2].max())[0]
    return problem.make_job(xyz, conn, z0, np.ones((len(z0), 3)), top,
                            np.tile([0.0, 10.0, 5.0], (len(top), 1)))


@pytest.mark.parametrize("k,rings", [(3, 2), (5, 2), (7, 3), (12, 1)])
def test_unstructured_mesh_parity(gpu_ctx, oracle, k, rings):
    """Irregular valence: 2k hexes meet at the centre line (k=7: 14 > 8 incidences -> second
    incidence chunk and the 128-candidate LDS sort; k=12: a 75-block row -> the wide-row
    accumulation path)."""
    job = _star_job(k, rings=rings)
    K, A = _assemble_both(gpu_ctx, oracle, job)
    rowptr, col, val = K.to_csr()
    assert np.array_equal(rowptr, A.ridx) and np.array_equal(col, A.idx)
    assert np.abs(val - A.vals).max() <= K_TOL * np.abs(A.vals).max()
    assert K.info()["max_row_blocks"] == (3 * (2 * k + 1) if rings == 1 else K.info()["max_row_blocks"])
    U, rep = K.cg_solve(job.F, 1e-12)
    Uo, repo = oracle.cg(A, job.F, 1e-12)
    assert np.abs(U - Uo).max() <= U_TOL * np.abs(Uo).max()
    K.free()


# (test_valence_limits_are_errors lived here until round 4: the limits it pinned -- 64 incidences, 96 blocks per row --
#  are gone; tests/test_gpu_assembly_edge.py assembles the same jobs and larger ones against the oracle)


def test_edge_cases_empty_and_fully_fixed(gpu_ctx, oracle):
    """Degenerate inputs go through without a crash: every DOF fixed (N = 0), a node that no
    element references (empty matrix row), and a model with nodes but no elements."""
    from stan_amd import host
    job = problem.cube_job(2)
    # every node clamped: the reduced system is empty, U is empty, termination type 1
    allfix = np.full(job.n_dof, -1, np.int32)
    K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                              job.mat_E_nu, allfix)
    assert K.info()["n_reduced"] == 0
    U, rep = K.cg_solve(np.zeros(0), 1e-8)
    assert U.shape == (0,) and rep["terminationtype"] == 1 and rep["iterations"] == 0
    rp, col, val = K.to_csr()
    assert rp.tolist() == [0] and col.size == 0
    K.free()
    # an orphan node appended after the mesh (DOFs numbered after the BFS range)
    xyz = np.vstack([job.xyz, [[9.0, 9.0, 9.0]]])
    nd = np.vstack([job.node_dof, [[81, 82, 83]]]).astype(np.int32)
    red, nfix = host.dof_reduction(84, nd, np.nonzero(job.xyz[:, 0] == 0)[0], np.ones((9, 3)))
    K = gpu_ctx.assemble_hex8(xyz, nd, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, red)
    rc, A = oracle.assemble(xyz, nd, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, red)
    rp, col, val = K.to_csr()
    assert np.array_equal(rp, A.ridx) and np.array_equal(col, A.idx)   # three empty rows at the end
    F = np.zeros(84 - nfix); F[:54] = job.F
    U, rep = K.cg_solve(F, 1e-10)
    Uo, _ = oracle.cg(A, F, 1e-10)
    assert np.abs(U - Uo).max() <= U_TOL * np.abs(Uo).max() and not U[54:].any()
    K.free()
    # nodes without elements: K = 0, CG reports "not positive definite" (p'Ap = 0), U = 0
    K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, np.zeros((0, 8), np.int32), np.zeros(0, np.int32),
                              np.zeros(0, np.uint8), job.mat_E_nu, job.red)
    assert K.info()["n_blocks"] == 0
    U, rep = K.cg_solve(job.F, 1e-8)
    assert rep["terminationtype"] == -5 and not U.any()
    K.free()


def test_overlap_option_does_not_change_the_bits(built_libs):
    import torch  # noqa: F401
    from stan_amd import hip
    job = problem.cube_job(9, jitter=0.1)
    args = (job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    res = []
    for overlap in (1, 0):
        ctx = hip.Context(0)
        ctx.comm_init(0, 1, ctx.unique_id())
        ctx.set_option(hip.OPT_OVERLAP_HALO, overlap)
        K = ctx.assemble_hex8(*args)
        res.append(K.cg_solve(job.F, 1e-10))
        K.free()
        ctx.close()
    assert res[0][1] == res[1][1] and np.array_equal(res[0][0], res[1][0])


@pytest.mark.parametrize("case", ["cube7", "mixed", "star7", "star12"])
def test_colour_scatter_assembly_mode_parity(built_libs, oracle, case):
    """STAN_OPT_ASSEMBLY_MODE = 1: one element per wavefront + colour-ordered scatter (the
    north-star variant) builds the same K as the default row-owner gather and as the oracle."""
    import torch  # noqa: F401
    from stan_amd import hip
    if case == "cube7":
        job = problem.cube_job(7, jitter=0.1)
    elif case == "mixed":
        job = problem.cube_job(6, jitter=0.1)
        rng = np.random.default_rng(5)
        job.elem_type = rng.integers(1, 3, job.conn.shape[0]).astype(np.uint8)
        job.elem_mat = rng.integers(0, 2, job.conn.shape[0]).astype(np.int32)
        job.mat_E_nu = np.array([[210000.0, 0.3], [70000.0, 0.33]])
    else:
        job = _star_job(int(case[4:]), rings=2 if case == "star7" else 1)
    ctx = hip.Context(0)
    ctx.set_profiling(True)
    ctx.set_option(hip.OPT_ASSEMBLY_MODE, 1)
    K, A = _assemble_both(ctx, oracle, job)
    ncol = ctx.profile()["assembly_colours"]
    assert 8 <= ncol <= 64
    rowptr, col, val = K.to_csr()
    assert np.array_equal(rowptr, A.ridx) and np.array_equal(col, A.idx)
    assert np.abs(val - A.vals).max() <= K_TOL * np.abs(A.vals).max()
    K2 = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                           job.mat_E_nu, job.red)
    assert np.array_equal(K2.to_csr()[2], val)            # deterministic
    U, rep = K.cg_solve(job.F, 1e-10)
    Uo, _ = oracle.cg(A, job.F, 1e-10)
    assert np.abs(U - Uo).max() <= 20 * U_TOL * np.abs(Uo).max()
    K.free(); K2.free(); ctx.close()


def test_fused_refresh_equals_literal_refresh(gpu_ctx, oracle):
    """Refresh iterations: r = b - (A x + a A p) from one matrix pass (default) against ALGLIB's
    literal second product b - A (x + a p): the same numbers up to rounding."""
    from stan_amd import hip
    job = problem.cube_job(14, jitter=0.05)
    K, A = _assemble_both(gpu_ctx, oracle, job)
    out = {}
    for fused in (1, 0):
        gpu_ctx.set_option(hip.OPT_CG_FUSED_REFRESH, fused)
        out[fused] = K.cg_solve(job.F, 1e-6)
    gpu_ctx.set_option(hip.OPT_CG_FUSED_REFRESH, 1)
    (U1, r1), (U0, r0) = out[1], out[0]
    Uo, repo = oracle.cg(A, job.F, 1e-6)
    assert r1["terminationtype"] == r0["terminationtype"] == repo["terminationtype"] == 1
    assert abs(r1["iterations"] - r0["iterations"]) <= 2 and abs(r0["iterations"] - repo["iterations"]) <= 3
    assert np.abs(U1 - U0).max() <= 1e-6 * np.abs(U0).max()
    assert np.abs(U1 - Uo).max() <= 1e-4 * np.abs(Uo).max()
    K.free()


def test_full_size_cube_properties(gpu_ctx):
    """BASELINE.json's headline size (148^3, 9.92 M DOF): the oracle cannot run here in seconds,
    so size-independent properties: block count formula, symmetry and linearity of the operator,
    CG to 1e-8 verified by an independent product, global equilibrium, fp32-matrix agreement."""
    from stan_amd import hip
    n = 148
    job = problem.cube_job(n)
    assert (job.n_dof, job.n_red) == (9923847, 9857244)
    K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                              job.mat_E_nu, job.red)
    info = K.info()
    assert info["n_blocks"] == (3 * n + 1) ** 3 == 88121125 and info["max_row_blocks"] == 27
    rng = np.random.default_rng(7)
    x, y = rng.standard_normal(job.n_red), rng.standard_normal(job.n_red)
    Kx, Ky = K.spmv(x), K.spmv(y)
    assert abs(y @ Kx - x @ Ky) <= 1e-9 * abs(y @ Kx)
    assert np.abs(K.spmv(0.5 * x + 2 * y) - (0.5 * Kx + 2 * Ky)).max() <= 1e-10 * np.abs(Kx).max()
    # rigid translation of the free DOFs is NOT in the null space (the clamp couples them), but a
    # constant vector produces forces only near the clamped face
    ones = K.spmv(np.ones(job.n_red))
    assert np.count_nonzero(np.abs(ones) > 1e-6 * np.abs(ones).max()) <= 3 * 2 * (n + 1) ** 2
    gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
    try:
        U, rep = K.cg_solve(job.F, 1e-8)
        Um, repm = K.cg_solve(job.F, 1e-8, precision_mode=hip.PREC_MIXED)
        Ux, repx = K.cg_solve(job.F, 1e-8, precision_mode=hip.PREC_FIXED48)
    finally:
        gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 1)
    assert rep["terminationtype"] == 1 and rep["rel_residual"] <= 1e-8
    # independent residual (unscaled system, separate kernel path): kappa(S) bounds the ratio
    K2 = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                               job.mat_E_nu, job.red)
    r = job.F - K2.spmv(U)
    assert np.linalg.norm(r) <= 1e-6 * np.linalg.norm(job.F)
    # equilibrium: the reactions at the clamp balance the applied load, so 1'K_free U = sum F
    # only up to the clamp's share; the z-load sum itself is exact
    assert np.isclose(job.F.sum(), 50.0 * (n + 1) ** 2)
    # cantilever under +z end load: every free node moves up, the tip most
    disp = np.zeros(job.n_dof); disp[job.red != -1] = U
    uz = disp[job.node_dof[:, 2]]
    assert uz.min() > -1e-9 * uz.max() and uz.argmax() in np.nonzero(job.xyz[:, 0] == n)[0]
    assert repm["terminationtype"] == 1
    # fp32 matrix entries perturb K by 6e-8 relative: the solution moves by up to kappa * 6e-8
    # (kappa of the scaled operator ~3e5 at this size)
    mixed_err = np.abs(Um - U).max() / np.abs(U).max()
    print("mixed-precision relative deviation at 148^3: %.2e" % mixed_err)
    assert mixed_err <= 2e-2
    # the 48-bit fixed-point stream perturbs entries by 7e-15 of the diagonal: same iteration
    # count, solution inside the north-star's 1e-6 with orders of magnitude to spare
    fx_err = np.abs(Ux - U).max() / np.abs(U).max()
    print("fixed-48 relative deviation at 148^3: %.2e (%d vs %d iterations)"
          % (fx_err, repx["iterations"], rep["iterations"]))
    assert repx["terminationtype"] == 1 and abs(repx["iterations"] - rep["iterations"]) <= 15
    assert fx_err <= 1e-7
    K.free(); K2.free()


def test_fuzz_random_jobs_against_oracle(gpu_ctx, oracle):
    """Random sub-meshes with shuffled wire order, two materials, G1/G2 mixes, partial SPCs and
    duplicate loads (tests/fuzz.py; tools/fuzz_parity.py runs the long sweep): pattern bit-exact,
    values to 1e-12, CG checked through an independent residual."""
    from tests import fuzz
    checked = 0
    for seed in range(40):
        job = fuzz.random_job(seed)
        if job is None:
            continue
        fuzz.check_job(gpu_ctx, oracle, job)
        checked += 1
    assert checked >= 25


def test_fuzz_shards_of_random_jobs(built_libs):
    """The partition / halo plan on shuffled, knocked-out meshes (row ranges far thinner than the
    BFS levels, neighbours beyond rank+-1, empty ranks): device plan == host plan, shard products
    bit-equal to the unsharded one."""
    import torch  # noqa: F401
    from stan_amd import hip
    from tests import fuzz
    checked = 0
    for seed in range(100, 124):
        job = fuzz.random_job(seed)
        if job is None:
            continue
        fuzz.check_shards(lambda: hip.Context(0), job, 2 + seed % 6)
        checked += 1
    assert checked >= 15


def test_more_ranks_than_slices(built_libs):
    """A 27-node mesh cut for 8 ranks: rank 7 owns the only slice, the others own nothing and
    must still assemble (empty shard) without an error."""
    import torch  # noqa: F401
    from stan_amd import hip, host
    job = problem.cube_job(2)
    ctx = hip.Context(0)
    owned = []
    for r in range(8):
        ctx.comm_init(r, 8, None)
        K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                              job.mat_E_nu, job.red)
        i = K.info()
        owned.append(i["row_end"] - i["row_begin"])
        ref = host.partition_plan(job.node_index, job.conn, 8, r)
        assert np.array_equal(K.plan()["halo_glob"], ref["halo_glob"]) and i["n_halo"] == 0
        K.free()
    assert sum(owned) == 27 and owned.count(0) == 7
    ctx.close()
