"""ctypes binding of libstan_host.so (include/stan_host.h): the host-side integer steps of
Solver.SolverLinearStatics around the GPU hot path.  No GPU needed."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libstan_host.so")

EXPORTS = ["stan_host_assign_dof", "stan_host_dof_reduction", "stan_host_load_vector",
           "stan_host_nodal_displacements"]

_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: run __graft_entry__.build()" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class StanHostError(RuntimeError):
    def __init__(self, code, what):
        super().__init__("libstan_host error %d in %s" % (code, what))
        self.code = code


def assign_dof(n_nodes, conn):
    """Database.AssignDOF (Database.cs:140-234) -> (node_index[n], node_dof[n,3])."""
    conn = np.ascontiguousarray(conn, dtype=np.int32).reshape(-1, 8)
    idx = np.zeros(n_nodes, dtype=np.int32)
    dof = np.zeros((n_nodes, 3), dtype=np.int32)
    rc = load().stan_host_assign_dof(C.c_int64(n_nodes), C.c_int64(conn.shape[0]),
                                     _p(conn, C.c_int32), _p(idx, C.c_int32), _p(dof, C.c_int32))
    if rc:
        raise StanHostError(rc, "assign_dof")
    return idx, dof


def dof_reduction(n_dof, node_dof, spc_nodes, spc_vals):
    """Solver.cs:104-132 -> (red[n_dof], n_fixed)."""
    node_dof = np.ascontiguousarray(node_dof, dtype=np.int32)
    spc_nodes = np.ascontiguousarray(spc_nodes, dtype=np.int32)
    spc_vals = np.ascontiguousarray(spc_vals, dtype=np.float64).reshape(-1, 3)
    red = np.zeros(n_dof, dtype=np.int32)
    nfix = C.c_int64(0)
    rc = load().stan_host_dof_reduction(C.c_int64(n_dof), _p(node_dof, C.c_int32),
                                        C.c_int64(spc_nodes.shape[0]), _p(spc_nodes, C.c_int32),
                                        _p(spc_vals, C.c_double), _p(red, C.c_int32),
                                        C.byref(nfix))
    if rc:
        raise StanHostError(rc, "dof_reduction")
    return red, int(nfix.value)


def load_vector(n_dof, node_dof, red, n_fixed, load_nodes, load_vals):
    """Solver.cs:136-152 -> F[n_dof - n_fixed]."""
    node_dof = np.ascontiguousarray(node_dof, dtype=np.int32)
    red = np.ascontiguousarray(red, dtype=np.int32)
    load_nodes = np.ascontiguousarray(load_nodes, dtype=np.int32)
    load_vals = np.ascontiguousarray(load_vals, dtype=np.float64).reshape(-1, 3)
    F = np.zeros(n_dof - n_fixed, dtype=np.float64)
    rc = load().stan_host_load_vector(C.c_int64(n_dof), _p(node_dof, C.c_int32),
                                      _p(red, C.c_int32), C.c_int64(load_nodes.shape[0]),
                                      _p(load_nodes, C.c_int32), _p(load_vals, C.c_double),
                                      _p(F, C.c_double))
    if rc:
        raise StanHostError(rc, "load_vector")
    return F


def nodal_displacements(node_dof, red, U):
    """Include_BC_DOF (SolverFunctions.cs:520-538) + Solver.cs:171-178 -> disp[n_nodes,3]."""
    node_dof = np.ascontiguousarray(node_dof, dtype=np.int32)
    red = np.ascontiguousarray(red, dtype=np.int32)
    U = np.ascontiguousarray(U, dtype=np.float64)
    out = np.zeros(node_dof.shape, dtype=np.float64)
    rc = load().stan_host_nodal_displacements(C.c_int64(node_dof.shape[0]),
                                              _p(node_dof, C.c_int32), _p(red, C.c_int32),
                                              _p(U, C.c_double), _p(out, C.c_double))
    if rc:
        raise StanHostError(rc, "nodal_displacements")
    return out
