"""ctypes binding of libstan_host.so (include/stan_host.h): the host-side integer steps of
Solver.SolverLinearStatics around the GPU hot path.  No GPU needed."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libstan_host.so")

EXPORTS = ["stan_host_assign_dof", "stan_host_dof_reduction", "stan_host_load_vector",
           "stan_host_nodal_displacements", "stan_host_partition_rows", "stan_host_partition_plan",
           "stan_host_cholesky_skyline_solve", "stan_host_lu_upper_solve", "stan_host_partition_elements"]

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


def cholesky_skyline_solve(rowptr, col, val, b):
    """LinearSolver_Cholesky (SolverFunctions.cs:332-444) on the reduced upper CRS: returns
    (x, terminationtype, profile_entries); terminationtype -3 and x = 0 when K is not SPD."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    n = rowptr.shape[0] - 1
    x = np.zeros(n)
    t, prof = C.c_int32(0), C.c_int64(0)
    rc = load().stan_host_cholesky_skyline_solve(C.c_int64(n), _p(rowptr, C.c_int64), _p(col, C.c_int32),
                                                 _p(val, C.c_double), _p(b, C.c_double), _p(x, C.c_double),
                                                 C.byref(t), C.byref(prof))
    if rc:
        raise StanHostError(rc, "cholesky_skyline_solve")
    return x, t.value, prof.value


def lu_upper_solve(rowptr, col, val, b):
    """LinearSolver_LU as the reference computes it (SolverFunctions.cs:446-516): triu(K) x = b."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    n = rowptr.shape[0] - 1
    x = np.zeros(n)
    t = C.c_int32(0)
    rc = load().stan_host_lu_upper_solve(C.c_int64(n), _p(rowptr, C.c_int64), _p(col, C.c_int32),
                                         _p(val, C.c_double), _p(b, C.c_double), _p(x, C.c_double), C.byref(t))
    if rc:
        raise StanHostError(rc, "lu_upper_solve")
    return x, t.value


def partition_plan(node_index, conn, nranks, rank):
    """Row partition + halo plan of `rank` (stan_host_partition_plan) as a dict."""
    node_index = np.ascontiguousarray(node_index, dtype=np.int32)
    conn = np.ascontiguousarray(conn, dtype=np.int32).reshape(-1, 8)
    n = node_index.shape[0]
    row_starts = np.zeros(nranks + 1, np.int64)
    halo = np.zeros(n, np.int32)
    send_rows = np.zeros(n, np.int32)
    nbr = np.zeros(nranks, np.int32)
    send_off = np.zeros(nranks + 1, np.int64)
    recv_off = np.zeros(nranks + 1, np.int64)
    nh, nn = C.c_int64(0), C.c_int32(0)
    rc = load().stan_host_partition_plan(
        C.c_int64(n), _p(node_index, C.c_int32), C.c_int64(conn.shape[0]), _p(conn, C.c_int32),
        C.c_int32(nranks), C.c_int32(rank), _p(row_starts, C.c_int64), C.byref(nh),
        _p(halo, C.c_int32), C.byref(nn), _p(nbr, C.c_int32), _p(send_off, C.c_int64),
        _p(send_rows, C.c_int32), _p(recv_off, C.c_int64))
    if rc:
        raise StanHostError(rc, "partition_plan")
    k = nn.value
    return dict(row_starts=row_starts, halo_glob=halo[:nh.value].copy(), nbr=nbr[:k].copy(),
                send_off=send_off[:k + 1].copy(), recv_off=recv_off[:k + 1].copy(),
                send_rows=send_rows[:send_off[k]].copy())


def partition_elements(node_index, conn, nranks, rank):
    """Ascending indices of the elements `rank` must hold (stan_host_partition_elements)."""
    node_index = np.ascontiguousarray(node_index, dtype=np.int32)
    conn = np.ascontiguousarray(conn, dtype=np.int32).reshape(-1, 8)
    out = np.zeros(conn.shape[0], np.int32)
    n = C.c_int64(0)
    rc = load().stan_host_partition_elements(C.c_int64(node_index.shape[0]), _p(node_index, C.c_int32),
                                             C.c_int64(conn.shape[0]), _p(conn, C.c_int32), C.c_int32(nranks),
                                             C.c_int32(rank), _p(out, C.c_int32), C.byref(n))
    if rc:
        raise StanHostError(rc, "partition_elements")
    return out[:n.value].copy()


# ---- STAN_Database mirror + STdb codec (stan_db part of include/stan_host.h) -----------------
EXPORTS += [
    "stan_host_db_new", "stan_host_db_free", "stan_host_db_last_error", "stan_host_db_read_stdb",
    "stan_host_db_parse_stdb", "stan_host_db_write_stdb", "stan_host_db_serialize",
    "stan_host_db_read_bdf", "stan_host_db_set_mesh", "stan_host_db_add_material",
    "stan_host_db_assign_part", "stan_host_db_add_bc", "stan_host_db_set_analysis",
    "stan_host_db_sizes", "stan_host_db_get_analysis", "stan_host_db_assign_dof",
    "stan_host_db_get_flat", "stan_host_db_get_reduction", "stan_host_db_set_results",
    "stan_host_db_get_results",
]


class Db:
    """A Database (Database.cs:10-21) living in libstan_host.so."""

    def __init__(self):
        self.lib = load()
        self.lib.stan_host_db_last_error.restype = C.c_char_p
        self.lib.stan_host_db_free.restype = None
        self.h = C.c_void_p()
        rc = self.lib.stan_host_db_new(C.byref(self.h))
        if rc:
            raise StanHostError(rc, "db_new")

    def __del__(self):
        try:
            if self.h:
                self.lib.stan_host_db_free(self.h)
                self.h = None
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc:
            msg = self.lib.stan_host_db_last_error(self.h)
            e = StanHostError(rc, what + ": " + (msg.decode() if msg else ""))
            raise e

    # -- STdb ---------------------------------------------------------------------------------
    @classmethod
    def read_stdb(cls, path):
        d = cls()
        d._chk(d.lib.stan_host_db_read_stdb(d.h, os.fsencode(path)), "read_stdb")
        return d

    @classmethod
    def parse_stdb(cls, data):
        d = cls()
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        d._chk(d.lib.stan_host_db_parse_stdb(d.h, buf, C.c_int64(len(data))), "parse_stdb")
        return d

    def write_stdb(self, path, packed=False):
        self._chk(self.lib.stan_host_db_write_stdb(self.h, os.fsencode(path), C.c_int32(int(packed))),
                  "write_stdb")

    def write_stdb_with_results(self, path, disp, strain, stress, packed=False):
        """ExportOutput straight from flat result arrays (same bytes as set_results + write_stdb)."""
        disp = np.ascontiguousarray(disp, dtype=np.float64)
        strain = np.ascontiguousarray(strain, dtype=np.float64)
        stress = np.ascontiguousarray(stress, dtype=np.float64)
        self._chk(self.lib.stan_host_db_write_stdb_with_results(
            self.h, os.fsencode(path), C.c_int32(int(packed)), _p(disp, C.c_double), _p(strain, C.c_double),
            _p(stress, C.c_double)), "write_stdb_with_results")

    def serialize(self, packed=False):
        n = C.c_int64(0)
        self._chk(self.lib.stan_host_db_serialize(self.h, C.c_int32(int(packed)), None, C.c_int64(0),
                                                  C.byref(n)), "serialize")
        buf = (C.c_uint8 * max(n.value, 1))()
        self._chk(self.lib.stan_host_db_serialize(self.h, C.c_int32(int(packed)), buf,
                                                  C.c_int64(n.value), C.byref(n)), "serialize")
        return bytes(buf[:n.value])

    # -- construction (what the GUI does before it launches the solver) -------------------------
    def read_bdf(self, path):
        nerr = C.c_int64(0)
        self._chk(self.lib.stan_host_db_read_bdf(self.h, os.fsencode(path), C.byref(nerr)), "read_bdf")
        return nerr.value

    def set_mesh(self, node_ids, xyz, elem_ids, elem_pids, nlist8, hex_type="HEX8_G2"):
        node_ids = np.ascontiguousarray(node_ids, dtype=np.int32)
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        elem_ids = np.ascontiguousarray(elem_ids, dtype=np.int32)
        elem_pids = np.ascontiguousarray(elem_pids, dtype=np.int32)
        nlist8 = np.ascontiguousarray(nlist8, dtype=np.int32)
        self._chk(self.lib.stan_host_db_set_mesh(
            self.h, C.c_int64(node_ids.shape[0]), _p(node_ids, C.c_int32), _p(xyz, C.c_double),
            C.c_int64(elem_ids.shape[0]), _p(elem_ids, C.c_int32), _p(elem_pids, C.c_int32),
            _p(nlist8, C.c_int32), hex_type.encode()), "set_mesh")

    def add_material(self, mid, name, E, nu):
        self._chk(self.lib.stan_host_db_add_material(self.h, C.c_int32(mid), name.encode(),
                                                     C.c_double(E), C.c_double(nu)), "add_material")

    def assign_part(self, pid, mat_id, hex_type="HEX8_G2"):
        self._chk(self.lib.stan_host_db_assign_part(self.h, C.c_int32(pid), C.c_int32(mat_id),
                                                    hex_type.encode()), "assign_part")

    def add_bc(self, bid, name, btype, node_ids, vals):
        node_ids = np.ascontiguousarray(node_ids, dtype=np.int32)
        vals = np.ascontiguousarray(vals, dtype=np.float64).reshape(-1, 3)
        self._chk(self.lib.stan_host_db_add_bc(self.h, C.c_int32(bid), name.encode(), btype.encode(),
                                               C.c_int64(node_ids.shape[0]), _p(node_ids, C.c_int32),
                                               _p(vals, C.c_double)), "add_bc")

    def set_analysis(self, atype="Linear_Statics", lin_solver="CG", tol=1e-6, max_iter=0, inc_numb=1):
        self._chk(self.lib.stan_host_db_set_analysis(self.h, atype.encode(), lin_solver.encode(),
                                                     C.c_double(tol), C.c_int32(max_iter),
                                                     C.c_int32(inc_numb)), "set_analysis")

    # -- queries ------------------------------------------------------------------------------
    def sizes(self):
        s = (C.c_int64 * 8)()
        self._chk(self.lib.stan_host_db_sizes(self.h, s), "sizes")
        return dict(nodes=s[0], elements=s[1], materials=s[2], bcs=s[3], nDOF=s[4],
                    result_step=s[5], import_errors=s[6])

    def analysis(self):
        t = C.create_string_buffer(64)
        ls = C.create_string_buffer(64)
        tol, mi, rs = C.c_double(0), C.c_int32(0), C.c_int32(0)
        self._chk(self.lib.stan_host_db_get_analysis(self.h, t, ls, C.c_int32(64), C.byref(tol),
                                                     C.byref(mi), C.byref(rs)), "get_analysis")
        return dict(type=t.value.decode(), lin_solver=ls.value.decode(), tol=tol.value,
                    max_iter=mi.value, result_step=rs.value)

    def assign_dof(self):
        self._chk(self.lib.stan_host_db_assign_dof(self.h), "assign_dof")

    def flat(self):
        s = self.sizes()
        n, e, nm = s["nodes"], s["elements"], max(s["materials"], 1)
        out = dict(xyz=np.zeros((n, 3)), node_ids=np.zeros(n, np.int32),
                   node_dof=np.zeros((n, 3), np.int32), conn=np.zeros((e, 8), np.int32),
                   elem_ids=np.zeros(e, np.int32), elem_mat=np.zeros(e, np.int32),
                   elem_type=np.zeros(e, np.uint8), mat_E_nu=np.zeros((nm, 2)))
        nmat = C.c_int32(0)
        self._chk(self.lib.stan_host_db_get_flat(
            self.h, _p(out["xyz"], C.c_double), _p(out["node_ids"], C.c_int32),
            _p(out["node_dof"], C.c_int32), _p(out["conn"], C.c_int32), _p(out["elem_ids"], C.c_int32),
            _p(out["elem_mat"], C.c_int32), _p(out["elem_type"], C.c_uint8),
            _p(out["mat_E_nu"], C.c_double), C.c_int32(nm), C.byref(nmat)), "get_flat")
        out["mat_E_nu"] = out["mat_E_nu"][:nmat.value]
        return out

    def reduction(self):
        ndof = self.sizes()["nDOF"]
        red = np.zeros(ndof, np.int32)
        F = np.zeros(max(ndof, 1))
        nfix = C.c_int64(0)
        self._chk(self.lib.stan_host_db_get_reduction(self.h, _p(red, C.c_int32), C.byref(nfix),
                                                      _p(F, C.c_double)), "get_reduction")
        return red, int(nfix.value), F[:ndof - nfix.value].copy()

    def set_results(self, disp, strain=None, stress=None):
        disp = np.ascontiguousarray(disp, dtype=np.float64)
        if strain is not None:
            strain = np.ascontiguousarray(strain, dtype=np.float64)
            stress = np.ascontiguousarray(stress, dtype=np.float64)
        self._chk(self.lib.stan_host_db_set_results(
            self.h, _p(disp, C.c_double), None if strain is None else _p(strain, C.c_double),
            None if stress is None else _p(stress, C.c_double)), "set_results")

    def results(self, inc=1, with_stress=True):
        s = self.sizes()
        disp = np.zeros((s["nodes"], 3))
        strain = np.zeros((s["elements"], 8, 6))
        stress = np.zeros((s["elements"], 8, 6))
        self._chk(self.lib.stan_host_db_get_results(
            self.h, C.c_int32(inc), _p(disp, C.c_double),
            _p(strain, C.c_double) if with_stress else None,
            _p(stress, C.c_double) if with_stress else None), "get_results")
        return disp, strain, stress
