"""Random linear-static jobs for the fuzz parity sweep (tests/test_gpu_parity.py::test_fuzz_* and
tools/fuzz_parity.py): a box of hexes with elements knocked out, node and element wire order
shuffled, jittered coordinates, two materials, HEX8_G1/G2 mixed, SPC entries with partial
components, random point loads.  Only meshes Database.AssignDOF accepts (connected) are kept."""
import numpy as np

from stan_amd import host, problem


def random_job(seed, collapse=0.0):
    """collapse > 0: that fraction of the elements becomes a wedge-shaped COLLAPSED hex (node 4 := node 1, node 8 :=
    node 5, as pre-processors write wedges into CHEXA cards): the element lists two nodes twice, which the reference
    accepts (Node.RemoveElemDuplicates, Node.cs:202-205) and scatters like any other K_e (SolverFunctions.cs:143-173)."""
    rng = np.random.default_rng(seed)
    nx, ny, nz = (int(v) for v in rng.integers(1, 8, 3))
    mx, my, mz = nx + 1, ny + 1, nz + 1
    k, j, i = np.meshgrid(np.arange(mz), np.arange(my), np.arange(mx), indexing="ij")
    xyz = np.stack([i.ravel(), j.ravel(), k.ravel()], axis=1).astype(np.float64)
    xyz *= rng.uniform(0.3, 3.0, 3)                       # anisotropic spacing
    xyz += rng.uniform(-0.12, 0.12, xyz.shape) * xyz.max(axis=0).clip(1e-9) / np.array([nx, ny, nz])

    def nid(a, b, c):
        return a + mx * (b + my * c)
    ke, je, ie = (v.ravel() for v in np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij"))
    conn = np.stack([nid(ie, je, ke), nid(ie + 1, je, ke), nid(ie + 1, je + 1, ke), nid(ie, je + 1, ke),
                     nid(ie, je, ke + 1), nid(ie + 1, je, ke + 1), nid(ie + 1, je + 1, ke + 1),
                     nid(ie, je + 1, ke + 1)], axis=1)
    keep = rng.random(conn.shape[0]) > rng.choice([0.0, 0.15, 0.4])
    if not keep.any():
        keep[rng.integers(conn.shape[0])] = True
    conn = conn[keep]
    conn = conn[rng.permutation(conn.shape[0])]          # element wire order
    used = np.unique(conn)
    perm = rng.permutation(used.shape[0])                 # node wire order
    new_of_old = np.full(xyz.shape[0], -1, dtype=np.int64)
    new_of_old[used] = perm
    xyz2 = np.empty((used.shape[0], 3))
    xyz2[perm] = xyz[used]
    conn = new_of_old[conn].astype(np.int32)
    n_nodes = used.shape[0]
    if collapse > 0:
        rc = np.random.default_rng(seed + 100003)     # (its own stream: collapse = 0 keeps every earlier job as it was)
        for e in np.nonzero(rc.random(conn.shape[0]) < collapse)[0]:
            old = conn[e].copy()
            conn[e, 3], conn[e, 7] = conn[e, 0], conn[e, 4]
            if np.unique(conn).shape[0] != n_nodes:   # would leave a node without an element: keep the hex
                conn[e] = old
    try:
        host.assign_dof(n_nodes, conn)
    except Exception:
        return None                                       # disconnected: the reference loops forever
    # BCs: clamp a random plane-ish subset fully, plus partial components elsewhere
    axis = int(rng.integers(3))
    lo = xyz2[:, axis] <= np.quantile(xyz2[:, axis], 0.15)
    spc_nodes = np.nonzero(lo)[0]
    spc_vals = np.ones((spc_nodes.shape[0], 3))
    extra = rng.choice(n_nodes, size=min(n_nodes, int(rng.integers(0, 6))), replace=False)
    extra_vals = rng.integers(0, 2, (extra.shape[0], 3)).astype(np.float64)
    extra_vals[rng.random(extra_vals.shape) < 0.1] = 2.0   # "== 1 exactly" filter (Solver.cs:110-112)
    spc_nodes = np.concatenate([spc_nodes, extra]).astype(np.int32)
    spc_vals = np.concatenate([spc_vals, extra_vals])
    ld = rng.choice(n_nodes, size=max(1, n_nodes // 5), replace=True).astype(np.int32)   # duplicates add up
    lv = rng.standard_normal((ld.shape[0], 3)) * 10.0
    job = problem.make_job(xyz2, conn, spc_nodes, spc_vals, ld, lv)
    job.elem_mat = rng.integers(0, 2, conn.shape[0]).astype(np.int32)
    job.mat_E_nu = np.array([[210000.0, 0.3], [float(rng.uniform(500, 70000)), float(rng.uniform(0.0, 0.42))]])
    g1 = rng.random() < 0.3
    job.elem_type = (rng.integers(1, 3, conn.shape[0]) if g1 else np.full(conn.shape[0], 2)).astype(np.uint8)
    job.has_g1 = bool(g1 and (job.elem_type == 1).any())
    return job


def check_job(ctx, oracle, job, cg=True):
    """Assembly pattern bit-exact, values to 1e-12, CG to the oracle's answer.  Returns a dict."""
    K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    rc, A = oracle.assemble(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    assert rc == 0
    out = {"n_red": int(job.n_red), "nnz": int(A.nnz)}
    if job.n_red == 0:
        K.free()
        return out
    rowptr, col, val = K.to_csr(upper_only=True)
    assert np.array_equal(rowptr, A.ridx) and np.array_equal(col, A.idx)
    out["k_err"] = float(np.abs(val - A.vals).max() / np.abs(A.vals).max())
    assert out["k_err"] <= 1e-12
    if cg and not job.has_g1:          # G1 elements leave hourglass modes: K may be singular
        import scipy.sparse as sp
        Au = sp.csr_matrix((A.vals, A.idx, A.ridx), shape=(A.n, A.n))
        Af = Au + sp.triu(Au, 1).T
        fn = np.linalg.norm(job.F)
        # merit stop off on both sides: the type-7 stop lands wherever rounding lets the merit
        # function tick up (residuals a decade apart between two correct runs); the residual
        # test is the comparable end state
        ctx.set_option(1, 0)   # STAN_OPT_CG_MERIT_STOP
        try:
            U, rep = K.cg_solve(job.F, 1e-9, 20000)
            U48, rep48 = K.cg_solve(job.F, 1e-9, 20000, precision_mode=2)
        finally:
            ctx.set_option(1, 1)
        Uo, repo = oracle.cg(A, job.F, 1e-9, 20000, merit_stop=False)
        out["term"] = (rep["terminationtype"], repo["terminationtype"])
        out["its"] = (rep["iterations"], repo["iterations"])
        # a floating sub-structure (no SPC reaches it) makes K singular: neither side converges
        # and the codes may differ; compare only where the oracle converged
        if repo["terminationtype"] == 1 and fn > 0:
            assert rep["terminationtype"] == 1, out
            if repo["iterations"] <= job.n_red:   # beyond N iterations CG runs on rounding noise
                assert abs(rep["iterations"] - repo["iterations"]) <= max(5, repo["iterations"] // 10), out
            out["res"] = (float(np.linalg.norm(job.F - Af @ U) / fn), float(np.linalg.norm(job.F - Af @ Uo) / fn))
            # independent residual in the UNSCALED norm; the stopping test is on the scaled one
            assert out["res"][0] <= max(10 * out["res"][1], 1e-7), out
            out["u_err"] = float(np.abs(U - Uo).max() / np.abs(Uo).max())
            assert out["u_err"] <= 1e-5, out                          # kappa * 1e-9, two runs
            assert rep48["terminationtype"] == 1, out
            out["res48"] = float(np.linalg.norm(job.F - Af @ U48) / fn)
            assert out["res48"] <= max(10 * out["res"][1], 1e-7), out
    K.free()
    return out


def check_shards(ctx_factory, job, nranks):
    """Detached ranks on one GPU: device halo plan == host plan, shard x [owned | halo] equals the
    rows of the unsharded product bit for bit, interior/boundary slice lists cover every slice."""
    from stan_amd import host
    args = (job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    ctx = ctx_factory()
    K1 = ctx.assemble_hex8(*args)
    x = np.random.default_rng(nranks).standard_normal(job.n_dof)
    y_ref = K1.spmv_local(x)
    K1.free()
    covered = 0
    for r in range(nranks):
        ctx.comm_init(r, nranks, None)
        K = ctx.assemble_hex8(*args)
        dev, ref = K.plan(), host.partition_plan(job.node_index, job.conn, nranks, r)
        assert np.array_equal(dev["row_starts"][:nranks + 1], ref["row_starts"])
        for k in ("halo_glob", "nbr", "send_off", "recv_off", "send_rows"):
            assert np.array_equal(dev[k], ref[k]), (k, r, nranks)
        r0, r1 = dev["row_begin"], dev["row_end"]
        xb = x.reshape(-1, 3)
        x_local = np.concatenate([xb[r0:r1], xb[dev["halo_glob"]]]).ravel()
        y = K.spmv_local(x_local)
        assert np.array_equal(y, y_ref[3 * r0:3 * r1]), (r, nranks)
        covered += r1 - r0
        K.free()
    assert covered == job.xyz.shape[0]
    ctx.close()


def random_revolved_job(seed):
    """A solid of revolution with collapsed hexes on its axis (stan_amd.cube.revolved_mesh), random sector / ring /
    layer counts (3 ... 160 sectors: up to 640 incidences and 483 blocks at an axis node), shuffled node and element wire
    order, two materials: the high-valence slow paths of the assembly (k_symbolic_big, chunked k_fill_cols,
    k_numeric_wide) and the duplicate-node branches on meshes no test wrote by hand."""
    from stan_amd.cube import revolved_mesh
    rng = np.random.default_rng(seed + 77000)
    sectors = int(rng.choice([3, 5, 8, 16, 17, 24, 33, 48, 72, 100, 160]))
    rings, layers = int(rng.integers(1, 4)), int(rng.integers(1, 5))
    xyz, conn = revolved_mesh(sectors, rings, layers, r0=float(rng.uniform(0.5, 2.0)), h=float(rng.uniform(0.5, 2.0)))
    n_nodes = xyz.shape[0]
    perm = rng.permutation(n_nodes)                       # node wire order
    xyz2 = np.empty_like(xyz)
    xyz2[perm] = xyz
    conn = perm[conn].astype(np.int32)
    conn = conn[rng.permutation(conn.shape[0])]           # element wire order
    zmin, zmax = xyz2[:, 2].min(), xyz2[:, 2].max()
    spc = np.nonzero(xyz2[:, 2] == zmin)[0].astype(np.int32)
    ld = np.nonzero(xyz2[:, 2] == zmax)[0].astype(np.int32)
    job = problem.make_job(xyz2, conn, spc, np.ones((spc.shape[0], 3)), ld, rng.standard_normal((ld.shape[0], 3)) * 10.0)
    job.elem_mat = rng.integers(0, 2, conn.shape[0]).astype(np.int32)
    job.mat_E_nu = np.array([[210000.0, 0.3], [float(rng.uniform(5000, 70000)), float(rng.uniform(0.0, 0.4))]])
    job.has_g1 = False
    job.sectors = sectors
    return job
