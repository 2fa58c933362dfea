"""CPU tests of libstan_host.so (host steps around the hot path) against the oracle."""
import numpy as np
import pytest

from stan_amd import host, problem
from stan_amd.cube import cube_bcs, cube_mesh


@pytest.mark.parametrize("n,jit", [(1, 0.0), (2, 0.0), (3, 0.1), (7, 0.0), (12, 0.1)])
def test_assign_dof_matches_literal_restatement(oracle, built_libs, n, jit):
    xyz, conn = cube_mesh(n, jitter=jit)
    idx, dof = host.assign_dof(xyz.shape[0], conn)
    rc, ref = oracle.assign_dof(xyz.shape[0], conn)
    assert rc == 0 and np.array_equal(idx, ref)          # bit-exact (int)
    assert sorted(idx.tolist()) == list(range(xyz.shape[0]))
    assert np.array_equal(dof, np.stack([3 * idx, 3 * idx + 1, 3 * idx + 2], 1))


def test_assign_dof_shuffled_elements_and_nodes(oracle, built_libs):
    # wire order drives the numbering (Dictionary iteration order): shuffle both
    rng = np.random.default_rng(5)
    xyz, conn = cube_mesh(5)
    perm = rng.permutation(xyz.shape[0])            # new position of each node
    inv = np.argsort(perm)
    conn2 = perm[conn][rng.permutation(conn.shape[0])].astype(np.int32)
    idx, _ = host.assign_dof(xyz.shape[0], conn2)
    rc, ref = oracle.assign_dof(xyz.shape[0], conn2)
    assert rc == 0 and np.array_equal(idx, ref)
    assert inv.shape == perm.shape


def test_assign_dof_degenerate_and_errors(oracle, built_libs):
    xyz, conn = cube_mesh(2)
    conn = conn.copy()
    conn[3, 7] = conn[3, 6]                           # an element listing a node twice
    idx, _ = host.assign_dof(xyz.shape[0], conn)
    rc, ref = oracle.assign_dof(xyz.shape[0], conn)
    assert rc == 0 and np.array_equal(idx, ref)
    with pytest.raises(host.StanHostError) as ei:      # two components
        host.assign_dof(54, np.concatenate([cube_mesh(2)[1], cube_mesh(2)[1] + 27]))
    assert ei.value.code == -21
    with pytest.raises(host.StanHostError) as ei:      # no start node
        host.assign_dof(4, np.zeros((0, 8), np.int32))
    assert ei.value.code == -20
    with pytest.raises(host.StanHostError) as ei:      # index out of range
        host.assign_dof(8, np.full((1, 8), 9, np.int32))
    assert ei.value.code == -2


def test_reduction_and_load_vector(oracle, built_libs):
    n = 3
    xyz, conn = cube_mesh(n)
    idx, dof = host.assign_dof(xyz.shape[0], conn)
    spc, ld, f = cube_bcs(n)
    vals = np.ones((spc.shape[0], 3))
    vals[0] = [1, 0, 1]          # partially fixed node
    vals[1] = [0.5, 2, 1]        # only exact 1 fixes (Solver.cs:110-112)
    spc2 = np.concatenate([spc, spc[:3]])            # duplicates are Distinct()-ed
    vals2 = np.concatenate([vals, vals[:3]])
    red, nfix = host.dof_reduction(3 * xyz.shape[0], dof, spc2, vals2)
    fixed = np.zeros(3 * xyz.shape[0], np.uint8)
    for s, v in zip(spc, vals):
        for d in range(3):
            if v[d] == 1:
                fixed[dof[s, d]] = 1
    nfix_ref, red_ref = oracle.dof_reduction(fixed)
    assert nfix == nfix_ref and np.array_equal(red, red_ref)
    # loads: duplicates accumulate, loads on fixed DOFs are dropped (Solver.cs:144)
    ln = np.concatenate([ld, ld[:2], spc[2:3]])
    lv = np.tile(f, (ln.shape[0], 1))
    F = host.load_vector(3 * xyz.shape[0], dof, red, nfix, ln, lv)
    Fref = np.zeros(3 * xyz.shape[0] - nfix)
    for nd, v in zip(ln, lv):
        for d in range(3):
            g = dof[nd, d]
            if red[g] != -1:
                Fref[g - red[g]] += v[d]
    assert np.array_equal(F, Fref)
    U = np.arange(F.shape[0], dtype=float) + 1
    disp = host.nodal_displacements(dof, red, U)
    assert np.array_equal(disp.ravel(), oracle.include_bc(red, U)[dof.ravel()])


def test_cube_job_sizes(built_libs):
    # SURVEY.md section 8 size table: N = nDOF - 3 (n+1)^2
    for n in (4, 10):
        j = problem.cube_job(n)
        assert j.n_dof == 3 * (n + 1) ** 3 and j.n_fixed == 3 * (n + 1) ** 2


@pytest.mark.parametrize("k", [3, 5, 7, 12])
def test_assign_dof_unstructured(oracle, built_libs, k):
    from stan_amd.cube import star_mesh
    xyz, conn = star_mesh(k, 3, 2)
    idx, _ = host.assign_dof(xyz.shape[0], conn)
    rc, ref = oracle.assign_dof(xyz.shape[0], conn)
    assert rc == 0 and np.array_equal(idx, ref)


def test_fuzz_host_steps_against_oracle(oracle, built_libs):
    """AssignDOF, nDOF_reduction and the load vector on random knocked-out, shuffled meshes
    (tests/fuzz.py) against the oracle's literal restatements: bit-exact integers, exact F."""
    from tests import fuzz
    checked = disconnected = 0
    for seed in range(200, 320):
        job = fuzz.random_job(seed)
        if job is None:
            disconnected += 1
            continue
        rc, ref = oracle.assign_dof(job.xyz.shape[0], job.conn)
        assert rc == 0 and np.array_equal(job.node_index, ref), seed
        # Solver.cs:121-132 restated: red[i] = -1 if fixed else number of fixed DOFs below i
        fixed = job.red == -1
        below = np.concatenate([[0], np.cumsum(fixed)[:-1]])
        assert np.array_equal(job.red[~fixed], below[~fixed]) and int(fixed.sum()) == job.n_fixed
        assert job.F.shape[0] == job.n_dof - job.n_fixed
        checked += 1
    assert checked >= 90 and disconnected >= 1


def test_fuzz_disconnected_mesh_is_an_error_on_both_sides(oracle, built_libs):
    """Where the reference's BFS runs off its list (Database.cs:218) both restatements refuse."""
    from tests import fuzz
    import numpy as np
    rng = np.random.default_rng(0)
    xyz, conn = cube_mesh(2)
    two = np.concatenate([conn, conn + xyz.shape[0]]).astype(np.int32)   # two separate cubes
    with pytest.raises(host.StanHostError):
        host.assign_dof(2 * xyz.shape[0], two)
    rc, _ = oracle.assign_dof(2 * xyz.shape[0], two)
    assert rc != 0
    assert fuzz.random_job(5) is None and rng is not None   # seed 5 of the sweep is such a mesh


def test_partition_elements_matches_brute_force(built_libs):
    """stan_host_partition_elements: the elements a rank must hold = those with a node in its rows."""
    from stan_amd import host, problem
    job = problem.cube_job(7, jitter=0.1)
    nb = job.xyz.shape[0]
    for nranks in (1, 2, 3, 5):
        seen = np.zeros(job.conn.shape[0], dtype=int)
        for rank in range(nranks):
            nsl = (nb + 63) // 64
            r0 = min(nb, nsl * rank // nranks * 64)
            r1 = nb if rank + 1 == nranks else min(nb, nsl * (rank + 1) // nranks * 64)
            rows = job.node_index[job.conn]
            want = np.nonzero(((rows >= r0) & (rows < r1)).any(axis=1))[0]
            got = host.partition_elements(job.node_index, job.conn, nranks, rank)
            assert np.array_equal(got, want.astype(np.int32))
            seen[got] += 1
        assert seen.min() >= 1 and (nranks == 1) == (seen.max() == 1)


def test_level_parallel_walk_gives_the_serial_numbering(built_libs, monkeypatch):
    """Round 5: AssignDOF's breadth-first walk scans wide levels on the host threads (claims by atomic minimum of
    (position in the level, offset in the scan): dof.cpp).  Database.cs:140-234's numbering is an ORDER, so the parallel
    form must reproduce the serial one exactly: cube, perforated box, a solid of revolution with a high-valence axis and
    forty fuzz meshes, every level forced through the threads (STAN_HOST_BFS_PAR_MIN=1) and with the default threshold."""
    from stan_amd.cube import cube_mesh, perforated_mesh, revolved_mesh
    from tests import fuzz
    meshes = {"cube": cube_mesh(24), "perforated": perforated_mesh(20, 0.4), "revolved": revolved_mesh(24, 6, 5)}
    for seed in range(100, 140):
        job = fuzz.random_job(seed)
        if job is not None:
            meshes["fuzz%d" % seed] = (job.xyz, job.conn)
    res = {}
    for tag, env in (("serial", {"STAN_HOST_THREADS": "1"}), ("forced", {"STAN_HOST_THREADS": "7", "STAN_HOST_BFS_PAR_MIN": "1"}),
                     ("default", {"STAN_HOST_THREADS": "8"})):
        with monkeypatch.context() as m:
            m.delenv("STAN_HOST_BFS_PAR_MIN", raising=False)
            for k, v in env.items():
                m.setenv(k, v)
            res[tag] = {name: host.assign_dof(xyz.shape[0], conn)[0] for name, (xyz, conn) in meshes.items()}
    for name in meshes:
        assert np.array_equal(res["serial"][name], res["forced"][name]), name
        assert np.array_equal(res["serial"][name], res["default"][name]), name
