"""A box of hexes with a fraction of its elements knocked out (largest connected component kept):
the irregular-mesh class of tests/fuzz.py at a chosen size.  Used by the SELL-C-sigma tests and tools."""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.csgraph

from stan_amd import problem
from stan_amd.cube import cube_mesh


def perforated_mesh(n, frac, seed=7):
    xyz, conn = cube_mesh(n)
    rng = np.random.default_rng(seed)
    conn = conn[rng.random(conn.shape[0]) >= frac]
    ne = conn.shape[0]
    inc = sp.csr_matrix((np.ones(ne * 8, dtype=np.int8), (np.repeat(np.arange(ne), 8), conn.ravel())),
                        shape=(ne, xyz.shape[0]))
    _, lab = sp.csgraph.connected_components((inc @ inc.T).tocsr(), directed=False)
    conn = conn[lab == np.argmax(np.bincount(lab))]
    used = np.unique(conn)
    new = np.full(xyz.shape[0], -1, dtype=np.int64)
    new[used] = np.arange(used.shape[0])
    return xyz[used], new[conn].astype(np.int32)


def perforated_job(n, frac, seed=7):
    """Clamp the nodes with x = 0, PointLoad (0,0,50) on the nodes with x = n (like the cube job)."""
    xyz, conn = perforated_mesh(n, frac, seed)
    spc = np.nonzero(xyz[:, 0] == 0.0)[0].astype(np.int32)
    ld = np.nonzero(xyz[:, 0] == float(n))[0].astype(np.int32)
    return problem.make_job(xyz, conn, spc, np.ones((spc.shape[0], 3)), ld,
                            np.tile(np.array([0.0, 0.0, 50.0]), (ld.shape[0], 1)))
