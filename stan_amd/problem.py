"""Linear-static job set-up in the flat-array form the C-ABI takes: the steps
Solver.Main / SolverLinearStatics perform before calling the hot path
(Solver.cs:46, :104-152), done by libstan_host.so."""
import numpy as np

from . import host
from .cube import cube_bcs, cube_mesh, perforated_mesh

HEX8_G1, HEX8_G2 = 1, 2


class Job:
    """xyz, conn, node_dof, red, F, elem_mat, elem_type, mat_E_nu, n_dof, n_fixed"""
    pass


def make_job(xyz, conn, spc_nodes, spc_vals, load_nodes, load_vals, etype=HEX8_G2,
             E=210000.0, nu=0.3):
    j = Job()
    j.xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    j.conn = np.ascontiguousarray(conn, dtype=np.int32)
    n_nodes = j.xyz.shape[0]
    j.node_index, j.node_dof = host.assign_dof(n_nodes, j.conn)          # Solver.cs:46
    j.n_dof = 3 * n_nodes                                                 # Database.cs:135-138
    j.red, j.n_fixed = host.dof_reduction(j.n_dof, j.node_dof, spc_nodes, spc_vals)
    j.F = host.load_vector(j.n_dof, j.node_dof, j.red, j.n_fixed, load_nodes, load_vals)
    j.elem_mat = np.zeros(j.conn.shape[0], dtype=np.int32)
    j.elem_type = np.full(j.conn.shape[0], etype, dtype=np.uint8)
    j.mat_E_nu = np.array([[E, nu]], dtype=np.float64)
    j.n_red = j.n_dof - j.n_fixed
    return j


def cube_job(n, etype=HEX8_G2, h=1.0, jitter=0.0, clamp_faces=None, E=210000.0, nu=0.3):
    """BASELINE.json's synthetic cube: clamp x=0 (G1: x=0,y=0,z=0 faces),
    PointLoad (0,0,50) on every node of the face x=n*h."""
    if clamp_faces is None:
        clamp_faces = "xyz" if etype == HEX8_G1 else "x"
    xyz, conn = cube_mesh(n, h=h, jitter=jitter)
    spc, ld, f = cube_bcs(n, h=h, clamp_faces=clamp_faces)
    spc_vals = np.ones((spc.shape[0], 3))
    load_vals = np.tile(f, (ld.shape[0], 1))
    return make_job(xyz, conn, spc, spc_vals, ld, load_vals, etype=etype, E=E, nu=nu)


def perforated_job(n, frac, seed=7, etype=HEX8_G2):
    """The cube job on cube.perforated_mesh: clamp the nodes with x = 0, PointLoad (0,0,50) on x = n."""
    xyz, conn = perforated_mesh(n, frac, seed)
    spc = np.nonzero(xyz[:, 0] == 0.0)[0].astype(np.int32)
    ld = np.nonzero(xyz[:, 0] == float(n))[0].astype(np.int32)
    return make_job(xyz, conn, spc, np.ones((spc.shape[0], 3)), ld,
                    np.tile(np.array([0.0, 0.0, 50.0]), (ld.shape[0], 1)), etype=etype)
