"""CPU fallback of the reference's direct-solver options (SURVEY.md section 8f rank 4;
SolverFunctions.cs:332-444 LinearSolver_Cholesky, :446-516 LinearSolver_LU) in libstan_host.so,
against scipy on the oracle's matrix (the reduced upper CRS alglib holds)."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from stan_amd import host, problem


def _system(oracle, n, etype=2, jitter=0.1):
    job = problem.cube_job(n, etype=etype, jitter=jitter)
    rc, A = oracle.assemble(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
    assert rc == 0
    U = sp.csr_matrix((A.vals, A.idx, A.ridx), shape=(A.n, A.n))
    return job, A, U


@pytest.mark.parametrize("n,etype", [(2, 2), (5, 2), (8, 2), (6, 1)])
def test_skyline_cholesky_against_scipy(built_libs, oracle, n, etype):
    job, A, U = _system(oracle, n, etype)
    K = (U + sp.triu(U, 1).T).tocsc()
    x, term, prof = host.cholesky_skyline_solve(A.ridx, A.idx, A.vals, job.F)
    xs = spl.spsolve(K, job.F)
    assert term == 1 and prof >= A.nnz
    assert np.abs(x - xs).max() <= 1e-9 * np.abs(xs).max()
    # and it is the CG's answer (the two LinSolver values solve the same system)
    Uo, _ = oracle.cg(A, job.F, 1e-12)
    assert np.abs(x - Uo).max() <= 1e-6 * np.abs(Uo).max()


def test_skyline_cholesky_reports_a_non_spd_matrix_like_alglib(built_libs, oracle):
    job, A, U = _system(oracle, 3)
    x, term, _ = host.cholesky_skyline_solve(A.ridx, A.idx, -A.vals, job.F)      # negative definite
    assert term == -3 and not x.any()                                            # "filled by zeros"
    with pytest.raises(host.StanHostError):                                      # a lower-triangle entry
        host.cholesky_skyline_solve(np.array([0, 1, 3]), np.array([0, 0, 1], dtype=np.int32), np.ones(3), np.ones(2))
    x, term, prof = host.cholesky_skyline_solve(np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0), np.zeros(0))
    assert term == 1 and prof == 0 and x.size == 0                               # empty system


def test_lu_solves_the_stored_triangle_like_the_reference(built_libs, oracle):
    """alglib.sparselu is handed the matrix ParallelAssembly_K built -- col >= row only
    (SolverFunctions.cs:158) -- so LinearSolver_LU returns the solution of triu(K) x = F."""
    job, A, U = _system(oracle, 4)
    x, term = host.lu_upper_solve(A.ridx, A.idx, A.vals, job.F)
    xs = spl.spsolve_triangular(U.tocsr(), job.F, lower=False)
    assert term == 1 and np.abs(x - xs).max() <= 1e-10 * np.abs(xs).max()
    K = (U + sp.triu(U, 1).T).tocsc()
    assert np.abs(x - spl.spsolve(K, job.F)).max() > 1e-3 * np.abs(x).max()      # NOT K's solution
