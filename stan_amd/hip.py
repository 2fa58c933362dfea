"""ctypes binding of libstan_hip.so (include/stan_hip.h).

This is plumbing for tests, bench.py and __graft_entry__: the product is the
shared library.  There is NO CPU fallback: loading fails loudly if the library
has not been built, and stan_hip_init fails if no GPU is present.

load() imports torch first (when installed) so that this library and torch share ONE HIP
runtime: torch bundles its own libamdhip64.so.7 and librccl.so.1.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STAN_HIP_LIB") or os.path.join(_HERE, "lib", "libstan_hip.so")

HEX8_G1, HEX8_G2 = 1, 2
PREC_FP64, PREC_MIXED, PREC_FIXED48 = 0, 1, 2
OPT_CG_MERIT_STOP, OPT_CG_RUPDATE, OPT_SPMV_VARIANT, OPT_OVERLAP_HALO, OPT_ASSEMBLY_MODE = 1, 2, 3, 4, 5
OPT_CG_FUSED_REFRESH = 6
OPT_POOL = 7
OPT_PLACEMENT_TRIES = 8
OPT_POOL_MAX_BYTES = 9
OPT_CG_SINGLE_REDUCE = 10
OPT_CG_FOLD_REDUCE = 11
OPT_VEC_STORE_NT = 12
OPT_PACKED_COLUMNS = 13
OPT_CG_DEFER_X = 14
OPT_SPMV_SMALL = 15
OPT_PLACEMENT_MAX_BYTES = 16
OPT_SELL_SIGMA = 17
OPT_COMM_P2P = 18
OPT_ROW_FOLDING = 19
OPT_CG_REFINE = 20
OPT_CG_LAZY_SCALING = 21
E_HIP, E_ARG, E_ALLOC, E_DETJ, E_DOF_LAYOUT, E_VALENCE, E_COMM, E_UNSUPPORTED = (
    -1, -2, -3, -4, -5, -6, -7, -8)

EXPORTS = [
    "stan_hip_init", "stan_hip_init_multi", "stan_hip_destroy", "stan_hip_last_error", "stan_hip_last_bad_element",
    "stan_hip_set_stream", "stan_hip_comm_unique_id", "stan_hip_comm_init",
    "stan_hip_assemble_hex8", "stan_hip_assemble_hex8_dev", "stan_hip_matrix_free",
    "stan_hip_cg_solve", "stan_hip_cg_solve_dev", "stan_hip_matrix_info", "stan_hip_ke_hex8",
    "stan_hip_ke_hex8_batch", "stan_hip_matrix_to_csr", "stan_hip_spmv", "stan_hip_spmv_bench", "stan_hip_stream_bench",
    "stan_hip_set_profiling", "stan_hip_get_profile", "stan_hip_set_option", "stan_hip_recover_hex8", "stan_hip_recover_hex8_dev",
    "stan_hip_nodal_forces_hex8", "stan_hip_pool_info",
    "stan_hip_matrix_plan", "stan_hip_spmv_local", "stan_hip_comm_info", "stan_hip_comm_library",
    "stan_hip_matrix_part_info", "stan_hip_get_profile_rank", "stan_hip_device_info", "stan_hip_matrix_diagonal",
    "stan_hip_recover_hex8_keep", "stan_hip_results_map", "stan_hip_results_free",
]
# only in the lab build (stan_amd/csrc/lab/stan_hip_lab.h, selected with STAN_HIP_LIB)
LAB_EXPORTS = ["stan_hip_csr_spmv_bench", "stan_hip_lab_placement_map", "stan_hip_lab_placement_variants", "stan_hip_lab_placement_alloc", "stan_hip_lab_placement_rounds", "stan_hip_lab_placement_cross", "stan_hip_lab_incg_penalty", "stan_hip_lab_placement_vecalloc", "stan_hip_lab_placement_vecshape", "stan_hip_lab_pairing_pmc"]


class MatrixInfo(C.Structure):
    _fields_ = [("n_dof", C.c_int64), ("n_reduced", C.c_int64), ("n_block_rows", C.c_int64),
                ("row_begin", C.c_int64), ("row_end", C.c_int64), ("n_halo", C.c_int64),
                ("n_blocks", C.c_int64), ("n_slots", C.c_int64), ("bytes_matrix", C.c_int64),
                ("scaled", C.c_int32), ("max_row_blocks", C.c_int32), ("n_elements_on_device", C.c_int64),
                ("sell_sigma", C.c_int32), ("folded_slots_permille", C.c_int32)]


class Profile(C.Structure):
    _fields_ = [("assemble_ms", C.c_double), ("symbolic_ms", C.c_double),
                ("numeric_ms", C.c_double), ("cg_ms", C.c_double), ("spmv_ms_total", C.c_double),
                ("spmv_launches", C.c_int64), ("spmv_bytes", C.c_int64),
                ("cg_iteration_vector_bytes", C.c_int64), ("iterations", C.c_int32),
                ("termination_type", C.c_int32), ("assembly_colours", C.c_int32),
                ("value_stream", C.c_int32), ("spmv2_ms_total", C.c_double), ("spmv2_launches", C.c_int64),
                ("loop_kernel_launches", C.c_int64), ("loop_collectives", C.c_int64),
                ("loop_iterations_enqueued", C.c_int64), ("placement_candidates", C.c_int32),
                ("placement_ms_best", C.c_float), ("placement_ms_worst", C.c_float),
                ("col_slots_packed", C.c_int64), ("placement_moved_vectors", C.c_int32), ("repacked_streams", C.c_int32),
                ("loop_stream_waits", C.c_int64), ("comm_reduce_ms_total", C.c_double),
                ("comm_reduce_calls", C.c_int64), ("comm_halo_ms_total", C.c_double), ("comm_halo_calls", C.c_int64),
                ("rel_residual_recurrence", C.c_double), ("rel_residual_fp64", C.c_double), ("refine_passes", C.c_int32),
                ("fp64_products", C.c_int32), ("fp64_products_ms", C.c_double)]


class StanHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libstan_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """dlopen the library (works without a GPU; compute calls do not)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % LIB_PATH)
        # torch bundles its own libamdhip64.so.7 / librccl.so.1.  If this library came first it
        # would pull in /opt/rocm's copies and a later `import torch` would add a SECOND HIP
        # runtime to the process (RCCL then fails with "no ROCm-capable device").  Loading
        # torch first makes both share one runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.stan_hip_last_error.restype = C.c_char_p
        _lib.stan_hip_last_bad_element.restype = C.c_int64
        _lib.stan_hip_destroy.restype = None
        _lib.stan_hip_matrix_free.restype = None
    return _lib


def _ptr(a, t):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(t))


def _dev(p, t):
    """device pointer given as int (torch tensor.data_ptr())"""
    return C.cast(C.c_void_p(int(p)), C.POINTER(t))


class Context:
    def __init__(self, device=0, devices=None):
        """device: one GPU.  devices=[...]: ONE handle driving several GPUs from this process
        (stan_hip_init_multi; ordinals may repeat only with the test transport tests/fake_rccl)."""
        self.lib = load()
        h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            rc = self.lib.stan_hip_init_multi(C.c_int(len(devices)), arr, C.byref(h))
        else:
            rc = self.lib.stan_hip_init(C.c_int(device), C.byref(h))
        if rc != 0:
            self.lib.stan_hip_last_error.argtypes = [C.c_void_p]
            msg = self.lib.stan_hip_last_error(None)
            raise StanHipError(rc, msg.decode() if msg else "stan_hip_init failed")
        self.h = h
        self.lib.stan_hip_last_error.argtypes = [C.c_void_p]
        self.lib.stan_hip_last_bad_element.argtypes = [C.c_void_p]
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.stan_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise StanHipError(rc, self.lib.stan_hip_last_error(self.h).decode())

    def last_bad_element(self):
        return int(self.lib.stan_hip_last_bad_element(self.h))

    def set_stream(self, stream_ptr):
        self._chk(self.lib.stan_hip_set_stream(self.h, C.c_void_p(stream_ptr)))

    def set_option(self, option, value):
        self._chk(self.lib.stan_hip_set_option(self.h, C.c_int32(option), C.c_int64(value)))

    def set_profiling(self, on=True):
        self._chk(self.lib.stan_hip_set_profiling(self.h, C.c_int32(1 if on else 0)))

    def profile(self):
        p = Profile()
        self._chk(self.lib.stan_hip_get_profile(self.h, C.byref(p)))
        return {k: getattr(p, k) for k, _ in Profile._fields_}

    def profile_rank(self, rank):
        """Profile of one rank of a multi-device handle (profile() reports rank 0's)."""
        p = Profile()
        self._chk(self.lib.stan_hip_get_profile_rank(self.h, C.c_int32(rank), C.byref(p)))
        return {k: getattr(p, k) for k, _ in Profile._fields_}

    def device_info(self, rank=0):
        """(HIP ordinal, PCI bus id) of the device a rank drives."""
        o, b = C.c_int32(-1), C.create_string_buffer(32)
        self._chk(self.lib.stan_hip_device_info(self.h, C.c_int32(rank), C.byref(o), b))
        return o.value, b.value.decode()

    def pool_info(self):
        """(bytes, blocks) of device memory the context keeps parked for reuse (OPT_POOL)."""
        b, n = C.c_int64(0), C.c_int64(0)
        self._chk(self.lib.stan_hip_pool_info(self.h, C.byref(b), C.byref(n)))
        return b.value, n.value

    # -- multi-GPU -----------------------------------------------------------------
    def unique_id(self):
        buf = C.create_string_buffer(128)
        rc = self.lib.stan_hip_comm_unique_id(buf)
        if rc != 0:
            raise StanHipError(rc, "ncclGetUniqueId failed")
        return bytes(buf.raw)

    def comm_init(self, rank, nranks, uid):
        """uid=None: detached rank (no collectives) for shard/plan checks."""
        if uid is None:
            self._chk(self.lib.stan_hip_comm_init(self.h, C.c_int(rank), C.c_int(nranks), None))
            return
        buf = C.create_string_buffer(bytes(uid), 128)
        self._chk(self.lib.stan_hip_comm_init(self.h, C.c_int(rank), C.c_int(nranks), buf))

    def comm_info(self):
        """What the sharded CG exchanges over: RCCL version code, the communicator's rank count / rank, p2p flag."""
        v, n, r, p = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
        self._chk(self.lib.stan_hip_comm_info(self.h, C.byref(v), C.byref(n), C.byref(r), C.byref(p)))
        buf, reused = C.create_string_buffer(4096), C.c_int32(0)
        self._chk(self.lib.stan_hip_comm_library(self.h, buf, C.c_int64(4096), C.byref(reused)))
        return dict(rccl_version=v.value, comm_ranks=n.value, comm_rank=r.value, p2p=bool(p.value),
                    library=buf.value.decode(), library_reused=bool(reused.value))

    # -- K_e (debug / parity) ---------------------------------------------------------
    def ke_hex8(self, xyz8, E, nu, etype):
        xyz8 = np.ascontiguousarray(xyz8, dtype=np.float64).reshape(24)
        out = np.zeros(576)
        self._chk(self.lib.stan_hip_ke_hex8(self.h, _ptr(xyz8, C.c_double), C.c_double(E),
                                            C.c_double(nu), C.c_int32(etype),
                                            _ptr(out, C.c_double)))
        return out.reshape(24, 24)

    def ke_hex8_batch(self, xyz8, E, nu, etypes):
        xyz8 = np.ascontiguousarray(xyz8, dtype=np.float64).reshape(-1, 24)
        n = xyz8.shape[0]
        etypes = np.ascontiguousarray(etypes, dtype=np.uint8)
        out = np.zeros((n, 576))
        self._chk(self.lib.stan_hip_ke_hex8_batch(self.h, C.c_int64(n), _ptr(xyz8, C.c_double),
                                                  C.c_double(E), C.c_double(nu),
                                                  _ptr(etypes, C.c_uint8), _ptr(out, C.c_double)))
        return out.reshape(n, 24, 24)

    # -- stress recovery ---------------------------------------------------------------
    def recover_hex8(self, xyz, disp, conn, elem_mat, elem_type, mat_E_nu):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        disp = np.ascontiguousarray(disp, dtype=np.float64)
        conn = np.ascontiguousarray(conn, dtype=np.int32).reshape(-1, 8)
        elem_mat = np.ascontiguousarray(elem_mat, dtype=np.int32)
        elem_type = np.ascontiguousarray(elem_type, dtype=np.uint8)
        mat_E_nu = np.ascontiguousarray(mat_E_nu, dtype=np.float64).reshape(-1, 2)
        ne = conn.shape[0]
        strain = np.zeros((ne, 8, 6))
        stress = np.zeros((ne, 8, 6))
        self._chk(self.lib.stan_hip_recover_hex8(
            self.h, C.c_int64(xyz.shape[0]), _ptr(xyz, C.c_double), _ptr(disp, C.c_double),
            C.c_int64(ne), _ptr(conn, C.c_int32), _ptr(elem_mat, C.c_int32),
            _ptr(elem_type, C.c_uint8), C.c_int32(mat_E_nu.shape[0]), _ptr(mat_E_nu, C.c_double),
            _ptr(strain, C.c_double), _ptr(stress, C.c_double)))
        return strain, stress

    def recover_hex8_keep(self, xyz, disp, conn, elem_mat, elem_type, mat_E_nu):
        """Stress recovery with the results kept on the device(s): returns a Results handle (map(e0, e1), free())."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        disp = np.ascontiguousarray(disp, dtype=np.float64)
        conn = np.ascontiguousarray(conn, dtype=np.int32).reshape(-1, 8)
        elem_mat = np.ascontiguousarray(elem_mat, dtype=np.int32)
        elem_type = np.ascontiguousarray(elem_type, dtype=np.uint8)
        mat_E_nu = np.ascontiguousarray(mat_E_nu, dtype=np.float64).reshape(-1, 2)
        h = C.c_void_p()
        self._chk(self.lib.stan_hip_recover_hex8_keep(
            self.h, C.c_int64(xyz.shape[0]), _ptr(xyz, C.c_double), _ptr(disp, C.c_double),
            C.c_int64(conn.shape[0]), _ptr(conn, C.c_int32), _ptr(elem_mat, C.c_int32),
            _ptr(elem_type, C.c_uint8), C.c_int32(mat_E_nu.shape[0]), _ptr(mat_E_nu, C.c_double), C.byref(h)))
        return Results(self, h, conn.shape[0])

    def nodal_forces_hex8(self, xyz, disp, node_dof, conn, elem_mat, elem_type, mat_E_nu):
        """Element.NodalForces [n_elem,24] and the assembled R [n_dof] (Solver.cs:184-196)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        disp = np.ascontiguousarray(disp, dtype=np.float64)
        node_dof = np.ascontiguousarray(node_dof, dtype=np.int32)
        conn = np.ascontiguousarray(conn, dtype=np.int32).reshape(-1, 8)
        elem_mat = np.ascontiguousarray(elem_mat, dtype=np.int32)
        elem_type = np.ascontiguousarray(elem_type, dtype=np.uint8)
        mat_E_nu = np.ascontiguousarray(mat_E_nu, dtype=np.float64).reshape(-1, 2)
        ne, n_dof = conn.shape[0], xyz.shape[0] * 3
        f = np.zeros((ne, 24))
        R = np.zeros(n_dof)
        self._chk(self.lib.stan_hip_nodal_forces_hex8(
            self.h, C.c_int64(xyz.shape[0]), _ptr(xyz, C.c_double), _ptr(disp, C.c_double),
            _ptr(node_dof, C.c_int32), C.c_int64(ne), _ptr(conn, C.c_int32), _ptr(elem_mat, C.c_int32),
            _ptr(elem_type, C.c_uint8), C.c_int32(mat_E_nu.shape[0]), _ptr(mat_E_nu, C.c_double),
            C.c_int64(n_dof), _ptr(f, C.c_double), _ptr(R, C.c_double)))
        return f, R

    # -- assembly ------------------------------------------------------------------------
    def assemble_hex8(self, xyz, node_dof, conn, elem_mat, elem_type, mat_E_nu, red):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64)
        node_dof = np.ascontiguousarray(node_dof, dtype=np.int32)
        conn = np.ascontiguousarray(conn, dtype=np.int32).reshape(-1, 8)
        elem_mat = np.ascontiguousarray(elem_mat, dtype=np.int32)
        elem_type = np.ascontiguousarray(elem_type, dtype=np.uint8)
        mat_E_nu = np.ascontiguousarray(mat_E_nu, dtype=np.float64).reshape(-1, 2)
        red = np.ascontiguousarray(red, dtype=np.int32)
        k = C.c_void_p()
        self._chk(self.lib.stan_hip_assemble_hex8(
            self.h, C.c_int64(xyz.shape[0]), _ptr(xyz, C.c_double), _ptr(node_dof, C.c_int32),
            C.c_int64(conn.shape[0]), _ptr(conn, C.c_int32), _ptr(elem_mat, C.c_int32),
            _ptr(elem_type, C.c_uint8), C.c_int32(mat_E_nu.shape[0]), _ptr(mat_E_nu, C.c_double),
            C.c_int64(red.shape[0]), _ptr(red, C.c_int32), C.byref(k)))
        return Matrix(self, k)

    def assemble_hex8_dev(self, n_nodes, d_xyz, d_node_dof, n_elem, d_conn, d_elem_mat,
                          d_elem_type, mat_E_nu, n_dof, d_red):
        """All d_* are device pointers (ints), e.g. torch tensor.data_ptr()."""
        mat_E_nu = np.ascontiguousarray(mat_E_nu, dtype=np.float64).reshape(-1, 2)
        k = C.c_void_p()
        self._chk(self.lib.stan_hip_assemble_hex8_dev(
            self.h, C.c_int64(n_nodes), _dev(d_xyz, C.c_double), _dev(d_node_dof, C.c_int32),
            C.c_int64(n_elem), _dev(d_conn, C.c_int32), _dev(d_elem_mat, C.c_int32),
            _dev(d_elem_type, C.c_uint8), C.c_int32(mat_E_nu.shape[0]), _ptr(mat_E_nu, C.c_double),
            C.c_int64(n_dof), _dev(d_red, C.c_int32), C.byref(k)))
        return Matrix(self, k)


class Results:
    """Strain / stress of a recovery kept on the device(s) (stan_results*)."""

    def __init__(self, ctx, handle, n_elem):
        self.ctx, self.h, self.n_elem = ctx, handle, n_elem
        ctx.lib.stan_hip_results_free.restype = None

    def map(self, e0, e1):
        """(strain, stress) of elements [e0, e1) as [e1 - e0, 8, 6] arrays (copies of the thread's staging memory)."""
        ps, pt = C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
        rc = self.ctx.lib.stan_hip_results_map(self.h, C.c_int64(e0), C.c_int64(e1), C.byref(ps), C.byref(pt))
        if rc != 0:
            raise StanHipError(rc, "stan_hip_results_map")
        n = (e1 - e0) * 48
        if n == 0:
            return np.zeros((0, 8, 6)), np.zeros((0, 8, 6))
        return (np.ctypeslib.as_array(ps, shape=(n,)).reshape(-1, 8, 6).copy(),
                np.ctypeslib.as_array(pt, shape=(n,)).reshape(-1, 8, 6).copy())

    def free(self):
        if getattr(self, "h", None):
            self.ctx.lib.stan_hip_results_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Matrix:
    """Opaque device-resident K (stan_matrix*)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.k = handle

    def free(self):
        if getattr(self, "k", None):
            self.ctx.lib.stan_hip_matrix_free(self.k)
            self.k = None

    def __del__(self):
        # valid for a matrix whose context is already closed too: stan_hip_destroy detaches its
        # matrices, and a detached matrix gives its buffers straight back to the driver
        try:
            self.free()
        except Exception:
            pass

    def info(self):
        i = MatrixInfo()
        self.ctx._chk(self.ctx.lib.stan_hip_matrix_info(self.k, C.byref(i)))
        return {k: getattr(i, k) for k, _ in MatrixInfo._fields_}

    def part_info(self, part):
        """info() of one shard of a matrix on a multi-device handle."""
        i = MatrixInfo()
        self.ctx._chk(self.ctx.lib.stan_hip_matrix_part_info(self.k, C.c_int32(part), C.byref(i)))
        return {k: getattr(i, k) for k, _ in MatrixInfo._fields_}

    def cg_solve(self, F, eps_f, max_its=0, precision_mode=PREC_FP64):
        F = np.ascontiguousarray(F, dtype=np.float64)
        U = np.zeros_like(F)
        term, its, rel = C.c_int32(0), C.c_int32(0), C.c_double(0)
        self.ctx._chk(self.ctx.lib.stan_hip_cg_solve(
            self.ctx.h, self.k, _ptr(F, C.c_double), C.c_double(eps_f), C.c_int32(max_its),
            C.c_int32(precision_mode), _ptr(U, C.c_double), C.byref(term), C.byref(its),
            C.byref(rel)))
        return U, dict(terminationtype=term.value, iterations=its.value, rel_residual=rel.value)

    def cg_solve_dev(self, d_F, d_U, eps_f, max_its=0, precision_mode=PREC_FP64):
        term, its, rel = C.c_int32(0), C.c_int32(0), C.c_double(0)
        self.ctx._chk(self.ctx.lib.stan_hip_cg_solve_dev(
            self.ctx.h, self.k, _dev(d_F, C.c_double), C.c_double(eps_f), C.c_int32(max_its),
            C.c_int32(precision_mode), _dev(d_U, C.c_double), C.byref(term), C.byref(its),
            C.byref(rel)))
        return dict(terminationtype=term.value, iterations=its.value, rel_residual=rel.value)

    def to_csr(self, upper_only=True):
        nnz = C.c_int64(0)
        lib, h = self.ctx.lib, self.ctx.h
        self.ctx._chk(lib.stan_hip_matrix_to_csr(h, self.k, C.c_int32(int(upper_only)),
                                                 C.byref(nnz), None, None, None))
        N = self.info()["n_reduced"]
        rowptr = np.zeros(N + 1, dtype=np.int64)
        col = np.zeros(max(nnz.value, 1), dtype=np.int32)
        val = np.zeros(max(nnz.value, 1), dtype=np.float64)
        self.ctx._chk(lib.stan_hip_matrix_to_csr(h, self.k, C.c_int32(int(upper_only)),
                                                 C.byref(nnz), _ptr(rowptr, C.c_int64),
                                                 _ptr(col, C.c_int32), _ptr(val, C.c_double)))
        return rowptr, col[:nnz.value], val[:nnz.value]

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros_like(x)
        self.ctx._chk(self.ctx.lib.stan_hip_spmv(self.ctx.h, self.k, _ptr(x, C.c_double),
                                                 _ptr(y, C.c_double)))
        return y

    def diagonal(self):
        """K_ii of the reduced system (unscaled)."""
        d = np.zeros(self.info()["n_reduced"])
        self.ctx._chk(self.ctx.lib.stan_hip_matrix_diagonal(self.ctx.h, self.k, _ptr(d, C.c_double)))
        return d

    def scaled_residual(self, F, U):
        """||S (F - K U)|| / ||S F|| with s_i = 1/sqrt(K_ii): the quantity stan_hip_cg_solve reports, from an
        independent product (stan_hip_spmv) and the exported diagonal."""
        d = self.diagonal()
        s = np.where(d > 0, 1.0 / np.sqrt(np.where(d > 0, d, 1.0)), 1.0)
        r = s * (np.asarray(F, dtype=np.float64) - self.spmv(U))
        return float(np.linalg.norm(r) / np.linalg.norm(s * F))

    def plan(self):
        lib, h = self.ctx.lib, self.ctx.h
        i = self.info()
        nr = 64
        row_starts = np.zeros(nr + 1, np.int64)
        nh, nn = C.c_int64(0), C.c_int32(0)
        self.ctx._chk(lib.stan_hip_matrix_plan(h, self.k, None, C.byref(nh), None, C.byref(nn),
                                               None, None, None, None))
        halo = np.zeros(max(nh.value, 1), np.int32)
        nbr = np.zeros(max(nn.value, 1), np.int32)
        send_off = np.zeros(nn.value + 1, np.int64)
        recv_off = np.zeros(nn.value + 1, np.int64)
        self.ctx._chk(lib.stan_hip_matrix_plan(h, self.k, _ptr(row_starts, C.c_int64), C.byref(nh),
                                               _ptr(halo, C.c_int32), C.byref(nn), _ptr(nbr, C.c_int32),
                                               _ptr(send_off, C.c_int64), None, _ptr(recv_off, C.c_int64)))
        send_rows = np.zeros(max(int(send_off[-1]), 1), np.int32)
        self.ctx._chk(lib.stan_hip_matrix_plan(h, self.k, None, C.byref(nh), None, C.byref(nn), None,
                                               None, _ptr(send_rows, C.c_int32), None))
        return dict(halo_glob=halo[:nh.value], nbr=nbr[:nn.value], send_off=send_off,
                    recv_off=recv_off, send_rows=send_rows[:int(send_off[-1])],
                    row_begin=i["row_begin"], row_end=i["row_end"], row_starts=row_starts)

    def spmv_local(self, x_local):
        i = self.info()
        x_local = np.ascontiguousarray(x_local, dtype=np.float64)
        assert x_local.shape[0] == 3 * (i["row_end"] - i["row_begin"] + i["n_halo"])
        y = np.zeros(3 * (i["row_end"] - i["row_begin"]))
        self.ctx._chk(self.ctx.lib.stan_hip_spmv_local(self.ctx.h, self.k, _ptr(x_local, C.c_double),
                                                       _ptr(y, C.c_double)))
        return y

    def csr_spmv_bench(self, reps=20):
        """lab build only (make -C stan_amd/csrc lab; STAN_HIP_LIB=.../libstan_hip_lab.so)"""
        if not hasattr(self.ctx.lib, "stan_hip_csr_spmv_bench"):
            raise RuntimeError("stan_hip_csr_spmv_bench exists only in the lab build of the library")
        ms, nb, diff = C.c_double(0), C.c_int64(0), C.c_double(0)
        self.ctx._chk(self.ctx.lib.stan_hip_csr_spmv_bench(self.ctx.h, self.k, C.c_int32(reps),
                                                           C.byref(ms), C.byref(nb), C.byref(diff)))
        return ms.value, nb.value, diff.value

    def stream_bench(self, reps=20):
        """(ms per read-only sweep of K's resident fp64 values in the product's access pattern, bytes per sweep)"""
        ms, nb = C.c_double(0), C.c_int64(0)
        self.ctx._chk(self.ctx.lib.stan_hip_stream_bench(self.ctx.h, self.k, C.c_int32(reps), C.byref(ms), C.byref(nb)))
        return ms.value, nb.value

    def spmv_bench(self, reps=20, precision_mode=PREC_FP64):
        ms = C.c_double(0)
        self.ctx._chk(self.ctx.lib.stan_hip_spmv_bench(self.ctx.h, self.k,
                                                       C.c_int32(precision_mode), C.c_int32(reps),
                                                       C.byref(ms)))
        return ms.value
