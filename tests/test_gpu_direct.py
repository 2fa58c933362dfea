"""LinSolver = "Cholesky" / "LU" through the console driver (Solver.cs:163-164): K is assembled
on the GPU, exported as the reduced upper CRS the reference's alglib.sparsematrix holds and solved
by the CPU fallback of libstan_host.so (stan_amd/host/direct.cpp)."""
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(tmp_path, n, solver):
    from stan_amd import host
    from stan_amd.cube import cube_bcs, cube_mesh
    xyz, conn = cube_mesh(n, jitter=0.1)
    d = host.Db()
    ne = conn.shape[0]
    d.set_mesh(np.arange(1, xyz.shape[0] + 1), xyz, np.arange(1, ne + 1), np.ones(ne), conn + 1, "HEX8_G2")
    d.add_material(1, "Steel", 210000.0, 0.3)
    d.assign_part(1, 1, "HEX8_G2")
    spc, ld, f = cube_bcs(n)
    d.add_bc(1, "fix", "SPC", spc + 1, np.ones((len(spc), 3)))
    d.add_bc(2, "load", "PointLoad", ld + 1, np.tile(f, (len(ld), 1)))
    d.set_analysis(lin_solver=solver, tol=1e-6)
    path = str(tmp_path / ("model_%s.STdb" % solver))
    d.write_stdb(path)
    return path


@pytest.mark.parametrize("solver", ["Cholesky", "LU"])
def test_console_driver_direct_solvers(built_libs, oracle, tmp_path, solver):
    from stan_amd import host
    exe = os.path.join(ROOT, "stan_amd", "bin", "stan_solver")
    path = _model(tmp_path, 6, solver)
    out = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert ("NORMAL termination" if solver == "Cholesky" else "NORMAL TERMINATION") in out.stdout
    r = host.Db.read_stdb(path)
    assert r.sizes()["result_step"] == 1
    disp = r.results(1)[0]
    m = host.Db.read_stdb(path); m.assign_dof()
    fl = m.flat(); red, nfix, F = m.reduction()
    rc, A = oracle.assemble(fl["xyz"], fl["node_dof"], fl["conn"], fl["elem_mat"], fl["elem_type"], fl["mat_E_nu"], red)
    Uu = sp.csr_matrix((A.vals, A.idx, A.ridx), shape=(A.n, A.n))
    if solver == "Cholesky":
        want = spl.spsolve((Uu + sp.triu(Uu, 1).T).tocsc(), F)
    else:   # the reference factorises the stored upper triangle (SolverFunctions.cs:158, 488)
        want = spl.spsolve_triangular(Uu.tocsr(), F, lower=False)
    do = host.nodal_displacements(fl["node_dof"], red, want)
    assert np.abs(disp - do).max() <= 1e-8 * np.abs(do).max()
