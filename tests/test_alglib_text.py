"""The ONLY alglib documentation inside /root/reference: the two ALGLIB-manual excerpts pasted as comments into
src/STAN_Solver/SolverFunctions.cs:278-298 (lincgsetcond) and :308-321 (lincgresults / Rep.TerminationType), plus the
reference's own reading of the codes at :323-329.  alglib.net 3.16.0 itself is a NuGet dependency that is not vendored
(src/STAN_Solver/packages.config:3), so the lincg restatement (oracle/stan_oracle.c, stan_amd/csrc/cg.hip) is from
memory -- EXCEPT for what these excerpts say.  One test per sentence, each quoting the line it pins, each run on both
sides: the CPU oracle (`oracle`, CPU suite) and the HIP path through the C-ABI (`hip`, GPU suite).

What the excerpts do NOT say (stays [recall]): the value of the "small" EpsF, that the residual
is the one of the diagonally scaled system, the refresh period, the merit rule behind type 7, restarts, and whether
"more than MaxIts" means > or >= (alglib's code stops at IterationsCount == MaxIts as far as recalled)."""
import numpy as np
import pytest
import scipy.sparse.linalg as sla

from stan_amd import problem

SIDES = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]


class _Solver:
    """solve(F, eps, maxits) -> (U, report) on one side; A = the oracle's CRS of the same job (for residual checks)."""

    def __init__(self, side, oracle, job, request):
        self.side, self.job = side, job
        rc, self.A = oracle.assemble(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)
        assert rc == 0
        self.oracle, self.K = oracle, None
        if side == "hip":
            ctx = request.getfixturevalue("gpu_ctx")
            self.K = ctx.assemble_hex8(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type, job.mat_E_nu, job.red)

    def solve(self, F, eps, maxits=0):
        if self.side == "hip":
            return self.K.cg_solve(F, eps, maxits)
        return self.oracle.cg(self.A, F, eps, maxits=maxits)

    def scaled_residual(self, F, U):
        """||S (F - K U)|| / ||S F|| with S = diag(K)^-1/2: the norm lincg's stopping test uses [recall: scaled]."""
        Kf = self.A.to_scipy_full().tocsr()
        s = 1.0 / np.sqrt(Kf.diagonal())
        return np.linalg.norm(s * (F - Kf @ U)) / np.linalg.norm(s * F)

    def close(self):
        if self.K is not None:
            self.K.free()


@pytest.fixture(params=SIDES)
def cg(request, oracle):
    made = []

    def make(job):
        made.append(_Solver(request.param, oracle, job, request))
        return made[-1]
    yield make
    for s in made:
        s.close()


def test_epsf_stops_when_the_residual_is_below_epsf_times_b(cg):
    """SolverFunctions.cs:284-285  "EpsF - algorithm will be stopped if norm of residual is less than EpsF*||b||."
    and :315  "* 1  ||residual||<=EpsF*||b||".  Type 1 means exactly that for the returned U (checked with an
    independent product), and one decade tighter costs more iterations."""
    s = cg(problem.cube_job(5, jitter=0.1))
    F = s.job.F
    its = []
    for eps in (1e-4, 1e-5, 1e-6):
        U, rep = s.solve(F, eps)
        assert rep["terminationtype"] == 1
        assert rep["rel_residual"] <= eps
        assert s.scaled_residual(F, U) <= eps * (1 + 1e-6)
        its.append(rep["iterations"])
    assert its[0] < its[1] < its[2]


def test_maxits_stops_the_iteration(cg):
    """SolverFunctions.cs:286-287  "MaxIts - algorithm will be stopped if number of iterations is more than MaxIts."
    and :316  "* 5  MaxIts steps was taken".  With an unreachable EpsF the run ends with type 5 and has taken MaxIts
    steps (never more); the point returned is the iterate of that step, not the start."""
    s = cg(problem.cube_job(4))
    for m in (1, 5, 12):
        U, rep = s.solve(s.job.F, 1e-30, maxits=m)
        assert rep["terminationtype"] == 5 and rep["iterations"] == m
        assert np.abs(U).max() > 0
    # (CG's residual norm may go up; its energy-norm error cannot)
    Kf = s.A.to_scipy_full().tocsc()
    Ud = sla.spsolve(Kf, s.job.F)
    err = [float((Ud - U) @ (Kf @ (Ud - U))) for U in (np.zeros_like(Ud), s.solve(s.job.F, 1e-30, maxits=5)[0],
                                                      s.solve(s.job.F, 1e-30, maxits=12)[0])]
    assert err[2] < err[1] < err[0]


def test_both_zero_means_a_small_epsf(cg):
    """SolverFunctions.cs:293-294  "If both EpsF and MaxIts are zero then small EpsF will be set to small value."
    (0, 0) neither runs forever nor stops at once: it ends with type 1 at a small residual; the value is 1e-6
    [recall: alglib's lincgcreate default], so the run is the bit-identical twin of an explicit (1e-6, 0).  The GUI's
    own defaults are (1e-6, 0) too (BOX_Analysis), so the reference never depends on the recalled value."""
    s = cg(problem.cube_job(4))
    U0, rep0 = s.solve(s.job.F, 0.0, 0)
    U1, rep1 = s.solve(s.job.F, 1e-6, 0)
    assert rep0["terminationtype"] == 1 and 0 < rep0["iterations"] and rep0["rel_residual"] <= 1e-6
    assert rep0["iterations"] == rep1["iterations"] and np.array_equal(U0, U1)
    # EpsF zero with MaxIts set is NOT that case: the cap rules
    _, rep = s.solve(s.job.F, 0.0, 3)
    assert rep["terminationtype"] == 5 and rep["iterations"] == 3


def test_minus_five_for_a_matrix_that_is_not_positive_definite(cg):
    """SolverFunctions.cs:311-312  "* -5  input matrix is either not positive definite, too large or too small".
    E < 0 makes K negative definite.  The reference prints ERROR for it (:323-324) and RETURNS U all the same (:329):
    here too the call succeeds and hands back a finite vector."""
    s = cg(problem.cube_job(3, E=-210000.0))
    U, rep = s.solve(s.job.F, 1e-8)
    assert rep["terminationtype"] == -5
    assert U.shape == s.job.F.shape and np.isfinite(U).all()


def test_minus_four_for_overflow_during_the_solution(cg):
    """SolverFunctions.cs:313-314  "* -4  overflow/underflow during solution (ill conditioned problem)".  A load of
    1e200 makes r.r overflow in the first step: reported as -4, not raised, U finite (the start point)."""
    s = cg(problem.cube_job(4))
    U, rep = s.solve(s.job.F * 1e200, 1e-8)
    assert rep["terminationtype"] == -4
    assert np.isfinite(U).all()


def test_seven_when_rounding_prevents_progress_and_the_best_point_is_returned(cg):
    """SolverFunctions.cs:317-318  "* 7  rounding errors prevent further progress, best point found is returned".
    An EpsF no fp64 run can meet ends with 7 -- the code the reference's own screenshot shows (images/Solver.PNG) and
    treats as NORMAL (:323) -- and the point handed back is as good as fp64 gets: the direct solver's answer to
    1e-7, with a residual many decades under the start's."""
    s = cg(problem.cube_job(4))
    U, rep = s.solve(s.job.F, 1e-30)
    assert rep["terminationtype"] == 7 and rep["iterations"] > 0
    Ud = sla.spsolve(s.A.to_scipy_full().tocsc(), s.job.F)
    assert np.abs(U - Ud).max() <= 1e-7 * np.abs(Ud).max()
    assert s.scaled_residual(s.job.F, U) <= 1e-6


def test_iterations_count_and_nmv(cg):
    """SolverFunctions.cs:319-320  "Rep.IterationsCount contains iterations count" / "NMV countains number of
    matrix-vector calculations".  The count is the number of CG steps taken (a zero right-hand side takes none); every
    step costs one product, so NMV >= IterationsCount (the oracle reports NMV; the C-ABI reports the count -- the
    reference reads neither, :323-329 -- and the library's profile counts the products it launched)."""
    s = cg(problem.cube_job(4))
    U, rep = s.solve(np.zeros_like(s.job.F), 1e-8)
    assert rep["terminationtype"] == 1 and rep["iterations"] == 0 and not U.any()
    U, rep = s.solve(s.job.F, 1e-6)
    assert rep["iterations"] > 0
    if s.side == "oracle":
        assert rep["nmv"] >= rep["iterations"]
        assert rep["nmv"] <= rep["iterations"] + rep["iterations"] // 10 + 2   # [recall] + a refresh product every 10 steps + r0
