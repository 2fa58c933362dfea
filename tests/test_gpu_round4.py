"""Round 4: degenerate elements and unlimited node valence on the GPU path, the bench-mode golden
fixtures of the sizes BASELINE.json names, config 5 at size."""
import os

import numpy as np
import pytest

from stan_amd import problem
from stan_amd.cube import cube_mesh, revolved_mesh

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K_TOL = 1e-13
OPT_ASSEMBLY_MODE = 5


def _job(xyz, conn, load=(0.0, 10.0, 5.0)):
    z0 = np.nonzero(xyz[:, 2] == xyz[:, 2].min())[0]
    top = np.nonzero(xyz[:, 2] == xyz[:, 2].max())[0]
    return problem.make_job(xyz, conn, z0, np.ones((len(z0), 3)), top, np.tile(load, (len(top), 1)))


def _check_against_oracle(ctx, oracle, job, mode, cg_eps=1e-10, u_tol=1e-6):
    ctx.set_option(OPT_ASSEMBLY_MODE, mode)
    try:
        K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    finally:
        ctx.set_option(OPT_ASSEMBLY_MODE, 0)
    rc, A = oracle.assemble(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    assert rc == 0
    rowptr, col, val = K.to_csr()
    assert np.array_equal(rowptr, A.ridx) and np.array_equal(col, A.idx)          # pattern: bit-equal
    kerr = np.abs(val - A.vals).max() / np.abs(A.vals).max()
    assert kerr <= K_TOL, kerr
    # merit-function stop off on both sides (its type-7 stop lands wherever rounding lets the merit tick up): the residual
    # test is the comparable end state; tolerance: |dU| <= u_tol max|U| (north-star bar 1e-6)
    ctx.set_option(1, 0)
    try:
        U, rep = K.cg_solve(job.F, cg_eps, 50000)
    finally:
        ctx.set_option(1, 1)
    Uo, repo = oracle.cg(A, job.F, cg_eps, 50000, merit_stop=False)
    assert rep["terminationtype"] == repo["terminationtype"] == 1, (rep, repo)
    assert abs(rep["iterations"] - repo["iterations"]) <= max(5, repo["iterations"] // 10), (rep, repo)
    assert np.abs(U - Uo).max() <= u_tol * np.abs(Uo).max()
    info = K.info()
    K.free()
    return info, kerr


@pytest.mark.parametrize("mode", [0, 1])
def test_wedge_collapsed_hex(gpu_ctx, oracle, mode):
    """A hex that lists a node twice (CHEXA wedge: node 4 := node 1, node 8 := node 5) in the middle of a 3^3 cube and
    on its corner: k_numeric's duplicate-node branch (assembly.hip) and k_scatter's (assembly_scatter.hip), which no
    GPU test fed before.  The reference accepts such elements (Node.cs:202-205)."""
    from stan_amd.cube import cube_bcs
    xyz, conn = cube_mesh(3, jitter=0.05)
    for e in (13, 0, 26):
        conn[e, 3], conn[e, 7] = conn[e, 0], conn[e, 4]
    spc, ld, f = cube_bcs(3)                 # (from the grid indices: the jittered coordinates have no exact planes)
    job = problem.make_job(xyz, conn, spc, np.ones((len(spc), 3)), ld, np.tile(f, (len(ld), 1)))
    info, _ = _check_against_oracle(gpu_ctx, oracle, job, mode)
    assert info["n_blocks"] > 0


@pytest.mark.parametrize("mode", [0, 1])
def test_revolved_mesh_with_72_collapsed_hexes_on_the_axis(gpu_ctx, oracle, mode):
    """VERDICT r03 missing 3: a solid of revolution in 5-degree sectors -- 72 collapsed hexes at every axis node, each
    listing it twice: 288 (element, local node) incidences (the fast symbolic kernel takes 64) and a row of 219 blocks
    (the fast numeric kernel's LDS takes 96).  The reference solves it (Database.cs:149-176, SolverFunctions.cs:143-173
    put no bound on the elements at a node); so does the library now: k_symbolic_big, k_fill_cols in chunks,
    k_numeric_wide; the colour scatter with as many colours as it takes."""
    xyz, conn = revolved_mesh(72, 2, 3)
    info, _ = _check_against_oracle(gpu_ctx, oracle, _job(xyz, conn), mode)
    assert info["max_row_blocks"] == 3 * (72 + 1)


@pytest.mark.parametrize("k,rings", [(20, 1), (33, 1), (40, 2), (130, 1)])
def test_star_meshes_beyond_the_old_limits(gpu_ctx, oracle, k, rings):
    """The jobs tests/test_gpu_parity.py used to expect STAN_E_VALENCE for (k = 20: a 123-block row; k = 33: 66
    incidences) and beyond (k = 40; k = 130: 260 incidences, 783 blocks): they assemble, pattern bit-equal, values to
    1e-13, and solve."""
    from stan_amd.cube import star_mesh
    xyz, conn = star_mesh(k, 3, rings)
    info, _ = _check_against_oracle(gpu_ctx, oracle, _job(xyz, conn), 0)
    if rings == 1:
        assert info["max_row_blocks"] == 3 * (2 * k + 1)


def test_a_node_with_thousands_of_incidences(gpu_ctx, oracle):
    """Beyond what the LDS sort of k_symbolic_big holds (3640 incidences): 1000 sectors x 2 layers x 2 listings = 4000
    at the interior axis node -- the global-scratch form of the same sort."""
    xyz, conn = revolved_mesh(1000, 1, 2)
    job = _job(xyz, conn)
    K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    rc, A = oracle.assemble(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    rowptr, col, val = K.to_csr()
    assert np.array_equal(rowptr, A.ridx) and np.array_equal(col, A.idx)
    assert np.abs(val - A.vals).max() <= 1e-12 * np.abs(A.vals).max()    # 4000 terms in one diagonal block
    assert K.info()["max_row_blocks"] == 3 * 1001
    x = np.random.default_rng(3).standard_normal(job.n_red)
    y = K.spmv(x)
    yo = A.to_scipy_full() @ x
    assert np.abs(y - yo).max() <= 1e-11 * np.abs(yo).max()
    K.free()


def test_fuzz_with_five_percent_collapsed_elements(gpu_ctx, oracle):
    """tests/fuzz.py jobs (shuffled wire order, knocked-out elements, two materials, partial SPCs) with 5 % of the
    elements collapsed into wedges: pattern bit-exact, values, CG against the oracle (fuzz.check_job), both modes'
    assembly."""
    from tests import fuzz
    done = 0
    for seed in range(300, 340):
        job = fuzz.random_job(seed, collapse=0.05)
        if job is None or job.has_g1:
            continue
        ncol = int((job.conn[:, 3] == job.conn[:, 0]).sum())
        if ncol == 0:
            continue
        fuzz.check_job(gpu_ctx, oracle, job)
        gpu_ctx.set_option(OPT_ASSEMBLY_MODE, 1)
        try:
            fuzz.check_job(gpu_ctx, oracle, job, cg=False)
        finally:
            gpu_ctx.set_option(OPT_ASSEMBLY_MODE, 0)
        done += 1
    assert done >= 10


def test_shards_of_the_revolved_mesh(built_libs):
    """The slow paths on a rank of a sharded run (halo discovery in k_symbolic_big): device plan == host plan, every
    shard reproduces its rows of the unsharded product bit for bit."""
    from stan_amd import hip
    from tests import fuzz
    xyz, conn = revolved_mesh(72, 2, 3)
    fuzz.check_shards(lambda: hip.Context(0), _job(xyz, conn), 3)


def test_two_base_packed_columns_at_a_size_with_wide_bfs_levels(gpu_ctx):
    """once a breadth-first level is wider than 2^16 rows, the slices that mix surface rows with
    27-neighbour rows no longer fit ONE 16-bit base per slot (63.9 % of the slots packed at 200^3, 98.4 % at 148^3);
    with a second base for the slice's shorter rows (k_pack_cols mode 2) practically all do.  160^3: levels up to 77 k
    rows wide.  Lossless: U, the iteration count and a plain product keep the bits of the int32 stream."""
    from stan_amd import hip
    job = problem.cube_job(160)
    K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    gpu_ctx.set_profiling(True)
    gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
    out = {}
    try:
        for packed in (1, 0):
            gpu_ctx.set_option(hip.OPT_PACKED_COLUMNS, packed)
            U, rep = K.cg_solve(job.F, 1e-8)
            pr = gpu_ctx.profile()
            x = np.random.default_rng(5).standard_normal(job.n_red)
            out[packed] = (U, rep, pr["col_slots_packed"], pr["spmv_bytes"], K.spmv(x))
    finally:
        gpu_ctx.set_option(hip.OPT_PACKED_COLUMNS, 1)
        gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 1)
        gpu_ctx.set_profiling(False)
    info = K.info()
    (Ua, ra, na, ba, ya), (Ub, rb, nb_, bb, yb) = out[1], out[0]
    assert ra == rb and ra["terminationtype"] == 1
    assert np.array_equal(Ua, Ub) and np.array_equal(ya, yb)
    assert nb_ == 0 and na >= 0.99 * info["n_slots"], (na, info["n_slots"])      # one base alone: ~0.93 here
    assert ba < bb
    K.free()


# ---- the sizes BASELINE.json names, against committed oracle fixtures ------------------------------
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _golden_job(name):
    if name.startswith("p"):
        n, frac = name[1:].split(":")
        return problem.perforated_job(int(n), float(frac)), "bench_mode_p%s_k%s.npz" % (n, frac)
    return problem.cube_job(int(name)), "bench_mode_%s.npz" % name


@pytest.mark.parametrize("name,fold", [("100", -1), ("148", -1), ("200", -1), ("p120:0.4", 0), ("p120:0.4", -1)])
def test_bench_mode_against_the_oracle_fixture(gpu_ctx, name, fold):
    """The oracle's answer on BASELINE.json's own sizes -- 100^3 (config 2), 148^3 (the headline), 200^3 (config 3)
    and the irregular 120^3 box (padded and folded streams) -- was computed once by
    tests/golden/make_bench_mode_golden.py (minutes to half an hour of CPU each) and committed: iteration count,
    termination type, U at a fixed sample of 4096 reduced DOFs, max|U|, sum U, ||U||.  The same job through the C-ABI,
    bench mode (merit stop off, eps 1e-8): iterations within 2, same code, max|dU| / max|U| <= 1e-9 (north-star bar
    1e-6; SolverFunctions.cs:270-330 is what these pin)."""
    from stan_amd import hip
    job, fname = _golden_job(name)
    path = os.path.join(GOLDEN, fname)
    if not os.path.exists(path):
        pytest.skip("fixture %s not generated yet" % fname)
    g = np.load(path)
    assert int(g["n_dof"]) == job.n_dof and int(g["n_red"]) == job.n_red and int(g["n_elem"]) == job.conn.shape[0]
    gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)
    gpu_ctx.set_option(hip.OPT_ROW_FOLDING, fold)
    gpu_ctx.set_profiling(True)
    try:
        K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
        U, rep = K.cg_solve(job.F, float(g["eps"]))
        folded = gpu_ctx.profile()["repacked_streams"]
        K.free()
    finally:
        gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 1)
        gpu_ctx.set_option(hip.OPT_ROW_FOLDING, -1)
        gpu_ctx.set_profiling(False)
    assert rep["terminationtype"] == int(g["terminationtype"]) == 1
    assert abs(rep["iterations"] - int(g["iterations"])) <= 2, (rep, int(g["iterations"]))
    um = float(g["u_max"])
    assert np.abs(U[g["idx"]] - g["U"]).max() <= 1e-9 * um
    assert abs(np.abs(U).max() - um) <= 1e-9 * um
    assert abs(U.sum() - float(g["u_sum"])) <= 1e-9 * um * np.sqrt(U.shape[0]) + 1e-9 * abs(float(g["u_sum"]))
    assert abs(np.sqrt(U @ U) - float(g["u_l2"])) <= 1e-9 * float(g["u_l2"])
    if name.startswith("p"):
        assert bool(folded) == (fold != 0)


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


def test_row_folding_always_reexamines_a_matrix_the_auto_rule_declined(gpu_ctx):
    """auto mode declines the cube (the plan saves nothing) and used to leave the matrix marked for
    good; STAN_OPT_ROW_FOLDING = 1 ("always") set afterwards must build the folded streams for that same matrix."""
    from stan_amd import hip
    job = problem.cube_job(20, jitter=0.05)
    gpu_ctx.set_option(hip.OPT_SPMV_SMALL, 0)          # (small systems never fold: take the large-system kernels)
    gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 0)       # (the type-7 stop lands wherever rounding lets the merit tick up)
    gpu_ctx.set_profiling(True)
    try:
        K = gpu_ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
        U0, rep0 = K.cg_solve(job.F, 1e-10)
        assert gpu_ctx.profile()["repacked_streams"] == 0
        gpu_ctx.set_option(hip.OPT_ROW_FOLDING, 1)
        U1, rep1 = K.cg_solve(job.F, 1e-10)
        assert gpu_ctx.profile()["repacked_streams"] == 1 and K.info()["folded_slots_permille"] > 0
        assert rep0["terminationtype"] == rep1["terminationtype"] and abs(rep0["iterations"] - rep1["iterations"]) <= 2
        assert np.abs(U1 - U0).max() <= 1e-8 * np.abs(U0).max()
        K.free()
    finally:
        gpu_ctx.set_option(hip.OPT_ROW_FOLDING, -1)
        gpu_ctx.set_option(hip.OPT_SPMV_SMALL, 1)
        gpu_ctx.set_option(hip.OPT_CG_MERIT_STOP, 1)
        gpu_ctx.set_profiling(False)


def test_console_driver_on_a_revolved_mesh(built_libs, oracle, tmp_path):
    """The whole console path (Solver.Main: STdb -> AssignDOF -> BC tables -> assembly -> CG -> stress recovery -> STdb) on
    a mesh with collapsed hexes and a high-valence axis: 36 sectors = 144 incidences at an axis node (the slow symbolic
    path) and a 111-block row (the wide numeric path); nodal displacements against a direct solve of the oracle's K."""
    import subprocess
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    from stan_amd import host
    xyz, conn = revolved_mesh(36, 2, 3)
    d = host.Db()
    ne = conn.shape[0]
    d.set_mesh(np.arange(1, xyz.shape[0] + 1), xyz, np.arange(1, ne + 1), np.ones(ne), conn + 1, "HEX8_G2")
    d.add_material(1, "Steel", 210000.0, 0.3)
    d.assign_part(1, 1, "HEX8_G2")
    z0 = np.nonzero(xyz[:, 2] == 0)[0]
    top = np.nonzero(xyz[:, 2] == xyz[:, 2].max())[0]
    d.add_bc(1, "fix", "SPC", z0 + 1, np.ones((len(z0), 3)))
    d.add_bc(2, "load", "PointLoad", top + 1, np.tile([0.0, 10.0, 5.0], (len(top), 1)))
    d.set_analysis(lin_solver="CG", tol=1e-10)
    path = str(tmp_path / "revolved.STdb")
    d.write_stdb(path)
    exe = os.path.join(ROOT, "stan_amd", "bin", "stan_solver")
    out = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    r = host.Db.read_stdb(path)
    assert r.sizes()["result_step"] == 1
    disp = r.results(1)[0]
    m = host.Db.read_stdb(path); m.assign_dof()
    fl = m.flat(); red, nfix, F = m.reduction()
    rc, A = oracle.assemble(fl["xyz"], fl["node_dof"], fl["conn"], fl["elem_mat"], fl["elem_type"], fl["mat_E_nu"], red)
    assert rc == 0
    Uu = sp.csr_matrix((A.vals, A.idx, A.ridx), shape=(A.n, A.n))
    want = spl.spsolve((Uu + sp.triu(Uu, 1).T).tocsc(), F)
    do = host.nodal_displacements(fl["node_dof"], red, want)
    assert np.abs(disp - do).max() <= 1e-5 * np.abs(do).max()      # the CG stops by its merit rule (type 7) near 1e-7


def test_stress_recovery_and_nodal_forces_on_collapsed_hexes(gpu_ctx, oracle):
    """Element.Recovery_Stress / Compute_NodalForces (Element.cs:211-255) on the revolved mesh: the wedge-collapsed hexes
    on the axis list a node twice -- their 8x6 strain / stress blocks (two rows then belong to the same node, each
    extrapolated with its own shape-function row, as the reference does) and the R assembly, where the repeated node
    receives both listings' forces, against the oracle element by element."""
    xyz, conn = revolved_mesh(24, 2, 2)
    job = _job(xyz, conn)
    disp = np.random.default_rng(11).standard_normal(job.xyz.shape) * 1e-3
    strain, stress = gpu_ctx.recover_hex8(job.xyz, disp, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu)
    f, R = gpu_ctx.nodal_forces_hex8(job.xyz, disp, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu)
    fo = np.zeros_like(f)
    E, nu = job.mat_E_nu[0]
    ncollapsed = 0
    for e in range(job.conn.shape[0]):
        rc, eo, so = oracle.recover_hex8(job.xyz[job.conn[e]], E, nu, 2, disp[job.conn[e]].ravel())
        rc2, fo[e] = oracle.nodal_forces_hex8(job.xyz[job.conn[e]], 2, so)
        assert rc == 0 and rc2 == 0
        assert np.abs(strain[e] - eo).max() <= 1e-11 * np.abs(eo).max()
        assert np.abs(stress[e] - so).max() <= 1e-11 * np.abs(so).max()
        ncollapsed += job.conn[e, 0] == job.conn[e, 3]
    assert ncollapsed == 48
    assert np.abs(f - fo).max() <= 1e-11 * np.abs(fo).max()
    Ro = np.zeros(job.n_dof)
    np.add.at(Ro, job.node_dof[job.conn].reshape(-1, 24), fo)
    assert np.abs(R - Ro).max() <= 1e-10 * np.abs(Ro).max()


def test_device_memory_does_not_grow_over_many_jobs():
    """A context that is created, used (both assembly modes, the high-valence symbolic path with its global scratch, fp64 /
    fp32-matrix / FIXED-48 solves, folded streams, recovery) and closed gives every byte back: 40 such lives leave the
    device's free memory where it was; inside ONE context 40 assemble / solve / free rounds stay within the block pool's
    bound (the pool parks blocks for reuse, it must not accumulate them)."""
    import torch
    from stan_amd import hip
    torch.cuda.synchronize()

    def free_mb():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info(0)[0] / 2 ** 20

    jobs = [problem.cube_job(12, jitter=0.1)]
    xyz, conn = revolved_mesh(72, 2, 2)
    jobs.append(_job(xyz, conn))
    xyz, conn = revolved_mesh(1000, 1, 2)          # a node with 4000 incidences: k_symbolic_big's global scratch
    jobs.append(_job(xyz, conn))

    def one_life():
        ctx = hip.Context(0)
        for i, job in enumerate(jobs):
            ctx.set_option(OPT_ASSEMBLY_MODE, i & 1)
            ctx.set_option(hip.OPT_ROW_FOLDING, 1 if i == 1 else -1)
            K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
            for prec in (hip.PREC_FP64, hip.PREC_MIXED, hip.PREC_FIXED48):
                U, rep = K.cg_solve(job.F, 1e-8, 400, prec)
            K.free()
        ctx.close()

    one_life()                                      # whatever the runtime keeps for itself is taken here
    one_life()
    before = free_mb()
    for _ in range(40):
        one_life()
    after = free_mb()
    assert before - after < 8, "40 context lives cost %.1f MB of device memory" % (before - after)

    ctx = hip.Context(0)
    marks = []
    for r in range(40):
        for job in jobs[:2]:
            K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
            K.cg_solve(job.F, 1e-8, 200)
            K.free()
        if r in (4, 39):
            marks.append(free_mb())
    ctx.close()
    assert marks[0] - marks[1] < 8, "rounds 5 -> 40 inside one context cost %.1f MB" % (marks[0] - marks[1])
    assert free_mb() >= before - 8


def test_placement_second_stage_moves_only_the_product_vectors(oracle):
    """placement.hip, second stage (round 4): when no candidate block is clear of the vectors' memory group, only the two
    vectors the products WRITE are re-allocated behind spacer blocks (tools/lab/spmv_steps_lab.cpp: the place of y alone
    decides 1.00 or 1.13 ms).  Whether the stage is entered and what it keeps depends on the box; here it is forced
    (STAN_PLACEMENT_TRACE=stage2): the solve that follows runs on the carved vectors and must give the bits of a solve without
    any search; the blocks are given back, the kept one with the context."""
    import torch
    from stan_amd import hip
    job = problem.cube_job(60)
    args = (job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    ctx = hip.Context(0)
    ctx.set_option(hip.OPT_PLACEMENT_TRIES, 1)
    K = ctx.assemble_hex8(*args)
    U0, rep0 = K.cg_solve(job.F, 1e-8)
    K.free()
    ctx.close()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    os.environ["STAN_PLACEMENT_TRACE"] = "stage2"      # enter the stage and adopt its best block whatever it gains
    try:
        ctx = hip.Context(0)
        ctx.set_option(hip.OPT_PLACEMENT_TRIES, 4)
        K = ctx.assemble_hex8(*args)
        prof = ctx.profile()
        U1, rep1 = K.cg_solve(job.F, 1e-8)
        K2 = ctx.assemble_hex8(*args)                  # a second matrix of the size: the parked block, no second search
        U2, rep2 = K2.cg_solve(job.F, 1e-8)
    finally:
        del os.environ["STAN_PLACEMENT_TRACE"]
    assert prof["placement_candidates"] >= 1 and prof["placement_moved_vectors"] == 2
    assert rep1 == rep0 and np.array_equal(U1, U0)
    assert rep2 == rep0 and np.array_equal(U2, U0)
    K.free(); K2.free()
    ctx.close()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(0)[0] >= free0 - (8 << 20)
