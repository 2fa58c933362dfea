"""Generates tests/golden/*.npz.

There are NO reference-produced vectors for this path (the reference has no tests and
cannot be built here), so these fixtures pin the oracle against INDEPENDENT results:
  * K_e known answers worked out analytically / with numpy.linalg,
  * displacements from scipy.sparse.linalg.spsolve (direct LU) on the oracle-assembled K.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import scipy.sparse.linalg as sla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from stan_amd import problem  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def direct_solution(job):
    rc, A = O.assemble(job.xyz, job.node_dof, job.conn, job.elem_mat, job.elem_type,
                       job.mat_E_nu, job.red)
    assert rc == 0
    U = sla.spsolve(A.to_scipy_full().tocsc(), job.F)
    return A, U


def main():
    out = {}
    # unit-cube element K_e (E=210000, nu=0.3): numpy restatement with np.linalg, fp64
    x = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0],
                  [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], float)
    for name, t in (("g2", 2), ("g1", 1)):
        rc, K = O.ke_hex8(x, 210000.0, 0.3, t)
        out["ke_unit_" + name] = K
    rng = np.random.default_rng(2024)
    xs = x * [1.3, 0.7, 2.1] + rng.uniform(-0.15, 0.15, (8, 3))
    out["ke_skew_xyz"] = xs
    for name, t in (("g2", 2), ("g1", 1)):
        rc, K = O.ke_hex8(xs, 70000.0, 0.33, t)
        out["ke_skew_" + name] = K
    # cubes: BFS numbering, reduction table, direct-solve displacements
    for n, jit, et in ((2, 0.0, 2), (4, 0.1, 2), (6, 0.0, 2), (5, 0.1, 1)):
        job = problem.cube_job(n, etype=et, jitter=jit)
        A, U = direct_solution(job)
        tag = "cube%d_%s_j%d" % (n, "g2" if et == 2 else "g1", int(jit * 10))
        out[tag + "_node_index"] = job.node_index
        out[tag + "_red"] = job.red
        out[tag + "_U"] = U
        out[tag + "_nnz_upper"] = np.array([A.nnz])
    np.savez_compressed(os.path.join(HERE, "hot_path_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
