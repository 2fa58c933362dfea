"""GPU parity tests added in round 2 (VERDICT r01 "next round" items 1, 4, 6 and the advisor's
medium finding): the exact bench mode under the oracle at 56^3, BASELINE configs 3 and 5, the
folded reductions and the single-reduction CG against the classic loop and the oracle.

Condition numbers quoted below are of the Jacobi-scaled operator S K S, measured in the build
container with scipy (eigsh, both ends) on the oracle's matrix: HEX8_G2 cube clamped on x=0:
kappa ~ 12.7 n^2 (51.3 / 117.9 / 202.5 for n = 2, 3, 4; 2.8e5 at 148);
HEX8_G1 cube clamped on x=0, y=0, z=0: 4.9e4 / 1.55e5 / 7.8e5 for n = 12 / 16 / 24, i.e.
kappa ~ 2.4 n^4 (no hourglass control: FE_Library.cs:63-89)."""
import numpy as np
import pytest

from stan_amd import problem

pytestmark = pytest.mark.gpu
U_TOL = 1e-6


def _assemble_both(ctx, oracle, job):
    K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                          job.mat_E_nu, job.red)
    rc, A = oracle.assemble(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                            job.mat_E_nu, job.red, n_threads=8)
    assert rc == 0
    return K, A


def test_spmv_variant_option_only_takes_kernels_that_compute_the_product(gpu_ctx):
    """The product library carries variants 0 / 9 / 12 (cg.hip); the A/B variants of round 1 --
    one of which returned wrong numbers on purpose -- exist in the lab build only."""
    from stan_amd import hip
    job = problem.cube_job(9, jitter=0.1)
    K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                              job.mat_E_nu, job.red)
    x = np.random.default_rng(1).standard_normal(job.n_red)
    ys = []
    try:
        for v in (0, 9, 12, -1):
            gpu_ctx.set_option(hip.OPT_SPMV_VARIANT, v)
            ys.append(K.spmv(x))
        for v in (1, 8, 13, 14, -2):
            with pytest.raises(hip.StanHipError) as ei:
                gpu_ctx.set_option(hip.OPT_SPMV_VARIANT, v)
            assert ei.value.code == hip.E_ARG
    finally:
        gpu_ctx.set_option(hip.OPT_SPMV_VARIANT, -1)
    for y in ys[1:3]:
        assert np.array_equal(y, ys[0])     # same arithmetic in the same order
    # auto on a system this small is the workgroup-per-slice kernel: the same products, summed in another order
    assert np.abs(ys[3] - ys[0]).max() <= 1e-13 * np.abs(ys[0]).max()
    K.free()


def test_mixed_solve_after_spmv_bench_on_a_fresh_matrix(gpu_ctx, oracle):
    """stan_hip_spmv_bench(MIXED) on an unscaled matrix left an fp32 copy of
    the UNSCALED K behind; the next MIXED solve scaled the fp64 values, kept the stale copy and
    returned S K^-1 S b with a small reported residual."""
    from stan_amd import hip
    job = problem.cube_job(10, jitter=0.05)
    K, A = _assemble_both(gpu_ctx, oracle, job)
    assert K.spmv_bench(2, hip.PREC_MIXED) > 0
    assert K.info()["scaled"] == 0
    U, rep = K.cg_solve(job.F, 1e-8, precision_mode=hip.PREC_MIXED)
    Uo, _ = oracle.cg(A, job.F, 1e-12)
    assert rep["terminationtype"] in (1, 7)
    assert np.abs(U - Uo).max() <= 1e-4 * np.abs(Uo).max()
    # and the other way round: FIXED-48 bench, then a FIXED-48 solve
    K2, _ = _assemble_both(gpu_ctx, oracle, job)
    assert K2.spmv_bench(2, hip.PREC_FIXED48) > 0
    U2, rep2 = K2.cg_solve(job.F, 1e-12, precision_mode=hip.PREC_FIXED48)
    assert np.abs(U2 - Uo).max() <= U_TOL * np.abs(Uo).max()
    K.free(); K2.free()


@pytest.mark.parametrize("n,prec", [(20, "fp64"), (40, "fp64"), (40, "fixed48"), (24, "mixed")])
def test_folded_reductions_give_the_bits_of_separate_reduction_launches(gpu_ctx, n, prec):
    """STAN_OPT_CG_FOLD_REDUCE: the last block of the producing kernel adds the partial sums in
    the order k_reduce uses: every scalar of every iteration, hence U and the iteration count,
    must be identical -- a stale partial read across XCDs would show up here."""
    from stan_amd import hip
    pm = {"fp64": hip.PREC_FP64, "fixed48": hip.PREC_FIXED48, "mixed": hip.PREC_MIXED}[prec]
    job = problem.cube_job(n, jitter=0.05)
    K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                              job.mat_E_nu, job.red)
    out = {}
    try:
        for fold in (1, 0, 1):
            gpu_ctx.set_option(hip.OPT_CG_FOLD_REDUCE, fold)
            out.setdefault(fold, []).append(K.cg_solve(job.F, 1e-10, precision_mode=pm))
    finally:
        gpu_ctx.set_option(hip.OPT_CG_FOLD_REDUCE, 1)
    (Ua, ra), (Uc, rc_) = out[1]
    Ub, rb = out[0][0]
    assert ra == rb == rc_ and ra["iterations"] > 50
    assert np.array_equal(Ua, Ub) and np.array_equal(Ua, Uc)
    K.free()


@pytest.mark.parametrize("n,etype,jit", [(10, 2, 0.05), (24, 2, 0.1), (12, 1, 0.05)])
def test_single_reduction_cg_against_classic_loop_and_oracle(gpu_ctx, oracle, n, etype, jit):
    """STAN_OPT_CG_SINGLE_REDUCE (Chronopoulos-Gear): same iterates in exact arithmetic, so the
    same U within the solver tolerance, the same termination code and an iteration count within
    2 % + 3 of the classic loop's (the oracle's)."""
    from stan_amd import hip
    job = problem.cube_job(n, etype=etype, jitter=jit)
    K, A = _assemble_both(gpu_ctx, oracle, job)
    gpu_ctx.set_profiling(True)
    try:
        for eps, merit in ((1e-12, 1), (1e-8, 0), (1e-6, 1)):
            gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, merit)
            res = {}
            for sr in (0, 1):
                gpu_ctx.set_option(hip.OPT_CG_SINGLE_REDUCE, sr)
                res[sr] = K.cg_solve(job.F, eps) + (gpu_ctx.profile(),)
            (U0, r0, p0), (U1, r1, p1) = res[0], res[1]
            Uo, repo = oracle.cg(A, job.F, eps, merit_stop=bool(merit))
            assert r1["terminationtype"] == r0["terminationtype"] == repo["terminationtype"]
            slack = max(3, repo["iterations"] // 50) if r0["terminationtype"] == 1 else max(5, repo["iterations"] // 4)
            assert abs(r1["iterations"] - r0["iterations"]) <= slack
            assert abs(r1["iterations"] - repo["iterations"]) <= slack
            tol = U_TOL if eps == 1e-12 else 1e-3
            if etype == 1:
                tol *= 20
            assert np.abs(U1 - U0).max() <= tol * np.abs(U0).max()
            assert np.abs(U1 - Uo).max() <= tol * np.abs(Uo).max()
            # launches per iteration: classic 3 (+1 on literal refreshes), single-reduction 2 (+2 on refreshes)
            k0, k1 = p0["loop_kernel_launches"] / p0["loop_iterations_enqueued"], \
                p1["loop_kernel_launches"] / p1["loop_iterations_enqueued"]
            assert 2.9 <= k0 <= 3.2 and 2.0 <= k1 <= 2.35, (k0, k1)
    finally:
        gpu_ctx.set_option(hip.OPT_CG_SINGLE_REDUCE, 0)
        gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 1)
        gpu_ctx.set_profiling(False)
    K.free()


def test_single_reduction_cg_termination_codes(gpu_ctx, oracle):
    from stan_amd import hip
    job = problem.cube_job(4)
    K, A = _assemble_both(gpu_ctx, oracle, job)
    gpu_ctx.set_option(hip.OPT_CG_SINGLE_REDUCE, 1)
    try:
        U, rep = K.cg_solve(job.F, 1e-30, max_its=5)
        Uo, repo = oracle.cg(A, job.F, 1e-30, maxits=5)
        assert rep["terminationtype"] == 5 and rep["iterations"] == 5
        assert np.abs(U - Uo).max() <= 1e-9 * np.abs(Uo).max()      # the same five iterates
        U, rep = K.cg_solve(job.F, 0.0, max_its=0)                   # both zero -> eps_f = 1e-6
        assert rep["terminationtype"] == 1 and rep["rel_residual"] <= 1e-6
        U, rep = K.cg_solve(job.F, 1e-30)                            # unreachable: type 7, best point
        assert rep["terminationtype"] == 7
        Ux, _ = oracle.cg(A, job.F, 1e-12)
        assert np.abs(U - Ux).max() <= U_TOL * np.abs(Ux).max()
        U, rep = K.cg_solve(np.zeros_like(job.F), 1e-8)
        assert rep["terminationtype"] == 1 and rep["iterations"] == 0 and not U.any()
        # not SPD: alglib's -5, U returned regardless
        jn = problem.cube_job(3, E=-210000.0)
        Kn = gpu_ctx.assemble_hex8(jn.xyz, jn.node_dof, jn.conn, jn.elem_mat, jn.elem_type, jn.mat_E_nu, jn.red)
        U, rep = Kn.cg_solve(jn.F, 1e-8)
        assert rep["terminationtype"] == -5
        Kn.free()
        # the fold switch does not change the bits of this loop either
        a = K.cg_solve(job.F, 1e-10)
        gpu_ctx.set_option(hip.OPT_CG_FOLD_REDUCE, 0)
        b = K.cg_solve(job.F, 1e-10)
        assert a[1] == b[1] and np.array_equal(a[0], b[0])
    finally:
        gpu_ctx.set_option(hip.OPT_CG_FOLD_REDUCE, 1)
        gpu_ctx.set_option(hip.OPT_CG_SINGLE_REDUCE, 0)
    K.free()


def test_bench_mode_against_the_oracle_at_56_cubed(gpu_ctx, oracle):
    """the EXACT configuration bench.py times -- merit-function stop off,
    eps 1e-8, fp64 and FIXED-48 streams -- against oracle.cg(merit_stop=False) on the 56^3 cube
    (555 579 DOF; the oracle needs ~10 s for it on the GPU box's host).
    kappa(S K S) ~ 12.7 * 56^2 = 4.0e4, so two solves stopped at ||r|| <= 1e-8 ||b|| may differ by
    up to kappa * eps = 4e-4 relative; tightened to 1e-12 they must agree to the north-star 1e-6."""
    from stan_amd import hip
    n = 56
    kappa = 12.7 * n * n
    job = problem.cube_job(n)
    K, A = _assemble_both(gpu_ctx, oracle, job)
    rowptr, col, val = K.to_csr(upper_only=True)
    assert np.array_equal(rowptr, A.ridx) and np.array_equal(col, A.idx)
    assert np.abs(val - A.vals).max() <= 1e-13 * np.abs(A.vals).max()
    gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
    try:
        Uo8, ro8 = oracle.cg(A, job.F, 1e-8, merit_stop=False)
        Uo12, ro12 = oracle.cg(A, job.F, 1e-12, merit_stop=False)
        assert ro8["terminationtype"] == ro12["terminationtype"] == 1
        for prec in (hip.PREC_FP64, hip.PREC_FIXED48):
            U8, r8 = K.cg_solve(job.F, 1e-8, precision_mode=prec)
            U12, r12 = K.cg_solve(job.F, 1e-12, precision_mode=prec)
            assert r8["terminationtype"] == r12["terminationtype"] == 1
            assert r8["rel_residual"] <= 1e-8 and r12["rel_residual"] <= 1e-12
            assert abs(r8["iterations"] - ro8["iterations"]) <= max(2, ro8["iterations"] // 50), (r8, ro8)
            if prec == hip.PREC_FP64:
                assert abs(r12["iterations"] - ro12["iterations"]) <= max(2, ro12["iterations"] // 50), (r12, ro12)
            else:   # 1e-12 is below what the quantised entries carry (7e-15 * kappa): the fp64 check asks for a refinement pass
                assert ro12["iterations"] <= r12["iterations"] <= 2 * ro12["iterations"], (r12, ro12)
            d8 = np.abs(U8 - Uo8).max() / np.abs(Uo8).max()
            d12 = np.abs(U12 - Uo12).max() / np.abs(Uo12).max()
            print("56^3 bench mode, value stream %d: its %d/%d (oracle %d/%d), |U-Uo| %.2e at 1e-8, %.2e at 1e-12"
                  % (prec, r8["iterations"], r12["iterations"], ro8["iterations"], ro12["iterations"], d8, d12))
            assert d8 <= kappa * 1e-8
            assert d12 <= U_TOL
    finally:
        gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 1)
    K.free()


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


@pytest.mark.parametrize("n,prec,jit", [(20, "fp64", 0.05), (33, "fp64", 0.0), (24, "fixed48", 0.1), (24, "mixed", 0.05)])
def test_packed_column_stream_gives_the_same_bits(gpu_ctx, n, prec, jit):
    """STAN_OPT_PACKED_COLUMNS: 16-bit column offsets from a per-slot base, two slots per dword --
    lossless, the same products in the same order: U, the iteration count and a plain product must
    be bit-identical to the int32 column stream; nearly every slot of a BFS-ordered mesh packs
    (the ragged last slice, padded with column 0, is one that may not)."""
    from stan_amd import hip
    pm = {"fp64": hip.PREC_FP64, "fixed48": hip.PREC_FIXED48, "mixed": hip.PREC_MIXED}[prec]
    job = problem.cube_job(n, jitter=jit)
    K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                              job.mat_E_nu, job.red)
    gpu_ctx.set_profiling(True)
    out = {}
    try:
        for packed in (1, 0, 1):
            gpu_ctx.set_option(hip.OPT_PACKED_COLUMNS, packed)
            U, rep = K.cg_solve(job.F, 1e-10, precision_mode=pm)
            pr = gpu_ctx.profile()
            x = np.random.default_rng(3).standard_normal(job.n_red)
            out.setdefault(packed, []).append((U, rep, pr["col_slots_packed"], pr["spmv_bytes"], K.spmv(x)))
    finally:
        gpu_ctx.set_option(hip.OPT_PACKED_COLUMNS, 1)
        gpu_ctx.set_profiling(False)
    (Ua, ra, na, ba, ya), (Uc, rc_, nc, bc, yc) = out[1]
    Ub, rb, nb_, bb, yb = out[0][0]
    assert ra == rb == rc_ and ra["iterations"] > 30
    assert np.array_equal(Ua, Ub) and np.array_equal(Ua, Uc) and np.array_equal(ya, yb)
    info = K.info()
    assert nb_ == 0 and na == nc and 0.9 * info["n_slots"] <= na <= info["n_slots"]
    assert ba < bb     # the profile prices the bytes of the stream that ran
    K.free()


@pytest.mark.parametrize("fused_refresh", [1, 0])
def test_deferred_x_update_gives_the_same_bits(gpu_ctx, oracle, fused_refresh):
    """STAN_OPT_CG_DEFER_X (merit stop off): x' = x + alpha p formed by k_update instead of k_step --
    the same operands in the same expression, so every stop (residual, MaxIts in the middle of a
    refresh cycle, right on a refresh iteration) must return the same bits; and the answer is the
    oracle's."""
    from stan_amd import hip
    job = problem.cube_job(18, jitter=0.05)
    K, A = _assemble_both(gpu_ctx, oracle, job)
    gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
    gpu_ctx.set_option(hip.OPT_CG_FUSED_REFRESH, fused_refresh)
    try:
        for eps, maxits in ((1e-9, 0), (1e-30, 7), (1e-30, 10), (1e-30, 31), (1e-4, 0)):
            res = {}
            for d in (1, 0):
                gpu_ctx.set_option(hip.OPT_CG_DEFER_X, d)
                res[d] = K.cg_solve(job.F, eps, max_its=maxits)
            (U1, r1), (U0, r0) = res[1], res[0]
            assert r1 == r0, (r1, r0)
            assert np.array_equal(U1, U0), (eps, maxits)
            Uo, repo = oracle.cg(A, job.F, eps, maxits=maxits, merit_stop=False)
            assert r1["terminationtype"] == repo["terminationtype"]
            assert abs(r1["iterations"] - repo["iterations"]) <= max(2, repo["iterations"] // 50)
            if maxits:
                assert np.abs(U1 - Uo).max() <= 1e-9 * np.abs(Uo).max()      # the same iterates
    finally:
        gpu_ctx.set_option(hip.OPT_CG_DEFER_X, 1)
        gpu_ctx.set_option(hip.OPT_CG_FUSED_REFRESH, 1)
        gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 1)
    K.free()


@pytest.mark.parametrize("n,prec", [(6, "fp64"), (14, "fp64"), (14, "fixed48"), (12, "mixed")])
def test_small_system_spmv_kernel(gpu_ctx, oracle, n, prec):
    """STAN_OPT_SPMV_SMALL: up to 150 000 block rows one WORKGROUP owns a slice (four wavefronts take
    every fourth slot, partial rows added in a fixed order).  Same products as the one-wavefront kernel
    to rounding (<= 1e-14 of the row scale), the oracle's answer, bit-reproducible, and the same bits
    with the packed and the int32 column stream."""
    from stan_amd import hip
    pm = {"fp64": hip.PREC_FP64, "fixed48": hip.PREC_FIXED48, "mixed": hip.PREC_MIXED}[prec]
    job = problem.cube_job(n, jitter=0.1)
    K, A = _assemble_both(gpu_ctx, oracle, job)
    x = np.random.default_rng(5).standard_normal(job.n_red)
    out = {}
    try:
        for small in (1, 0):
            gpu_ctx.set_option(hip.OPT_SPMV_SMALL, small)
            Ka = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
            y = Ka.spmv(x)
            Ua, ra = Ka.cg_solve(job.F, 1e-10, precision_mode=pm)
            Ub, rb = Ka.cg_solve(job.F, 1e-10, precision_mode=pm)
            assert ra == rb and np.array_equal(Ua, Ub)                       # bit-reproducible
            gpu_ctx.set_option(hip.OPT_PACKED_COLUMNS, 0)
            Uc, rc_ = Ka.cg_solve(job.F, 1e-10, precision_mode=pm)
            gpu_ctx.set_option(hip.OPT_PACKED_COLUMNS, 1)
            assert ra == rc_ and np.array_equal(Ua, Uc)                      # same bits with int32 columns
            out[small] = (y, Ua, ra)
            Ka.free()
    finally:
        gpu_ctx.set_option(hip.OPT_SPMV_SMALL, 1)
        gpu_ctx.set_option(hip.OPT_PACKED_COLUMNS, 1)
    yo = oracle.smv_upper(A, x)
    assert np.abs(out[1][0] - yo).max() <= 1e-12 * np.abs(yo).max()
    assert np.abs(out[1][0] - out[0][0]).max() <= 1e-13 * np.abs(yo).max()
    Uo, repo = oracle.cg(A, job.F, 1e-10)
    tol = {"fp64": 1e-6, "fixed48": 1e-6, "mixed": 1e-3}[prec]
    for small in (1, 0):
        assert out[small][2]["terminationtype"] == repo["terminationtype"] or prec == "mixed"
        assert np.abs(out[small][1] - Uo).max() <= tol * np.abs(Uo).max()
    # (the fp32 matrix with eps 1e-10 ends on the merit-function floor, type 7: where exactly is rounding's choice)
    assert abs(out[1][2]["iterations"] - out[0][2]["iterations"]) <= max(3, out[0][2]["iterations"] // (10 if prec == "mixed" else 20))
    K.free()
