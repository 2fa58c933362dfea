"""ctypes binding of oracle/libstan_oracle.so.

TEST INFRASTRUCTURE ONLY (see oracle/stan_oracle.h): importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never from stan_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libstan_oracle.so")

HEX8_G1, HEX8_G2 = 1, 2


class _CRS(C.Structure):
    _fields_ = [("n", C.c_int64), ("nnz", C.c_int64), ("ridx", C.POINTER(C.c_int64)),
                ("idx", C.POINTER(C.c_int32)), ("vals", C.POINTER(C.c_double))]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
                os.path.join(_HERE, "stan_oracle.c")):
            build()
        _lib = C.CDLL(_SO)
        _lib.stan_oracle_dof_reduction.restype = C.c_int64
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def dn_dlocal(etype, g):
    out = np.zeros(24)
    n = lib().stan_oracle_dn_dlocal(etype, g, _p(out, C.c_double))
    assert n > 0
    return out.reshape(3, 8)


def extrap_N(etype):
    out = np.zeros(64)
    n = lib().stan_oracle_extrap_N(etype, _p(out, C.c_double))
    return out[: n * 8].reshape(n, 8)


def material_D(E, nu):
    D = np.zeros(36)
    lib().stan_oracle_material_D(C.c_double(E), C.c_double(nu), _p(D, C.c_double))
    return D.reshape(6, 6)


def ke_hex8(xyz8, E, nu, etype):
    xyz8 = np.ascontiguousarray(xyz8, dtype=np.float64).reshape(24)
    D = np.ascontiguousarray(material_D(E, nu)).reshape(36)
    K = np.zeros(576)
    rc = lib().stan_oracle_ke_hex8(_p(xyz8, C.c_double), _p(D, C.c_double), etype,
                                   _p(K, C.c_double))
    return rc, K.reshape(24, 24)


def recover_hex8(xyz8, E, nu, etype, dU):
    xyz8 = np.ascontiguousarray(xyz8, dtype=np.float64).reshape(24)
    dU = np.ascontiguousarray(dU, dtype=np.float64).reshape(24)
    D = np.ascontiguousarray(material_D(E, nu)).reshape(36)
    e = np.zeros(48)
    s = np.zeros(48)
    rc = lib().stan_oracle_recover_hex8(_p(xyz8, C.c_double), _p(D, C.c_double), etype,
                                        _p(dU, C.c_double), _p(e, C.c_double), _p(s, C.c_double))
    return rc, e.reshape(8, 6), s.reshape(8, 6)


def nodal_forces_hex8(xyz8, etype, stress_nodes):
    xyz8 = np.ascontiguousarray(xyz8, dtype=np.float64).reshape(24)
    sn = np.ascontiguousarray(stress_nodes, dtype=np.float64).reshape(48)
    f = np.zeros(24)
    rc = lib().stan_oracle_nodal_forces_hex8(_p(xyz8, C.c_double), etype, _p(sn, C.c_double),
                                             _p(f, C.c_double))
    return rc, f


def assign_dof(n_nodes, conn):
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    out = np.full(n_nodes, -1, dtype=np.int32)
    rc = lib().stan_oracle_assign_dof(C.c_int64(n_nodes), C.c_int64(conn.shape[0]),
                                      _p(conn, C.c_int32), _p(out, C.c_int32))
    return rc, out


def dof_reduction(fixed):
    fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
    red = np.zeros(fixed.shape[0], dtype=np.int32)
    nfix = lib().stan_oracle_dof_reduction(C.c_int64(fixed.shape[0]), _p(fixed, C.c_uint8),
                                           _p(red, C.c_int32))
    return int(nfix), red


class CRS:
    """Owns a stan_oracle_crs; exposes numpy copies."""

    def __init__(self, raw):
        self._raw = raw
        n, nnz = raw.n, raw.nnz
        self.n, self.nnz = n, nnz
        self.ridx = np.ctypeslib.as_array(raw.ridx, shape=(n + 1,)).copy()
        self.idx = np.ctypeslib.as_array(raw.idx, shape=(max(nnz, 1),))[:nnz].copy()
        self.vals = np.ctypeslib.as_array(raw.vals, shape=(max(nnz, 1),))[:nnz].copy()

    def __del__(self):
        try:
            lib().stan_oracle_crs_free(C.byref(self._raw))
        except Exception:
            pass

    def to_scipy_full(self):
        import scipy.sparse as sp
        U = sp.csr_matrix((self.vals, self.idx, self.ridx), shape=(self.n, self.n))
        return (U + sp.triu(U, 1).T).tocsr()


def assemble(xyz, node_dof, conn, elem_mat, elem_type, mat_E_nu, red, n_threads=1):
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    node_dof = np.ascontiguousarray(node_dof, dtype=np.int32)
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    elem_mat = np.ascontiguousarray(elem_mat, dtype=np.int32)
    elem_type = np.ascontiguousarray(elem_type, dtype=np.uint8)
    mat_E_nu = np.ascontiguousarray(mat_E_nu, dtype=np.float64)
    red = np.ascontiguousarray(red, dtype=np.int32)
    raw = _CRS()
    bad = C.c_int64(-1)
    rc = lib().stan_oracle_assemble(
        C.c_int64(xyz.shape[0]), _p(xyz, C.c_double), _p(node_dof, C.c_int32),
        C.c_int64(conn.shape[0]), _p(conn, C.c_int32), _p(elem_mat, C.c_int32),
        _p(elem_type, C.c_uint8), C.c_int32(mat_E_nu.shape[0]), _p(mat_E_nu, C.c_double),
        C.c_int64(red.shape[0]), _p(red, C.c_int32), C.c_int(n_threads), C.byref(raw),
        C.byref(bad))
    if rc != 0:
        return rc, bad.value
    return 0, CRS(raw)


def smv_upper(crs, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros(crs.n)
    lib().stan_oracle_smv_upper(C.byref(crs._raw), _p(x, C.c_double), _p(y, C.c_double))
    return y


def cg(crs, b, epsf, maxits=0, merit_stop=True, rupdate=10):
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(crs.n)
    term, its, nmv = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    rel = C.c_double(0)
    lib().stan_oracle_cg_opt(C.byref(crs._raw), _p(b, C.c_double), C.c_double(epsf),
                             C.c_int32(maxits), C.c_int(1 if merit_stop else 0), C.c_int(rupdate),
                             _p(x, C.c_double), C.byref(term), C.byref(its), C.byref(nmv),
                             C.byref(rel))
    return x, dict(terminationtype=term.value, iterations=its.value, nmv=nmv.value,
                   rel_residual=rel.value)


def set_mv_threads(n):
    lib().stan_oracle_set_mv_threads(C.c_int(int(n)))


def include_bc(red, U):
    red = np.ascontiguousarray(red, dtype=np.int32)
    U = np.ascontiguousarray(U, dtype=np.float64)
    out = np.zeros(red.shape[0])
    lib().stan_oracle_include_bc(C.c_int64(red.shape[0]), _p(red, C.c_int32), _p(U, C.c_double),
                                 _p(out, C.c_double))
    return out
