// api.hip -- the extern "C" surface of libstan_hip.so declared in include/stan_hip.h.
#include <cstring>

#include "internal.h"


namespace {
thread_local std::string g_err;  // errors raised before a context exists

template <typename T>
struct dbuf {  // RAII device buffer filled from host memory
    T *p = nullptr;
    stan_ctx *owner = nullptr;
    ~dbuf() { stan_dfree(owner, p); }
    int upload(stan_ctx *ctx, const T *h, size_t n) {
        owner = ctx;
        STANCHK(stan_dmalloc(ctx, &p, n));
        if (n) HIPCHK(ctx, hipMemcpyAsync(p, h, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        return STAN_OK;
    }
    int alloc(stan_ctx *ctx, size_t n) { owner = ctx; return stan_dmalloc(ctx, &p, n); }
};
}  // namespace

void stan_set_global_error(const std::string &msg) { g_err = msg; }

extern "C" {

int stan_hip_init(int device, stan_ctx **out) {
    if (!out) return STAN_E_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0 || device < 0 || device >= n) {
        g_err = std::string("stan_hip_init: no usable HIP device (") +
                (e != hipSuccess ? hipGetErrorString(e) : "device ordinal out of range") + ")";
        return STAN_E_HIP;
    }
    stan_ctx *c = new stan_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc((void **)&c->h_status, 64 * sizeof(int64_t), hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void **)&c->d_status, 64 * sizeof(int64_t)) != hipSuccess) {
        g_err = "stan_hip_init: stream / status allocation failed";
        delete c;
        return STAN_E_HIP;
    }
    c->own_stream = true;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) c->pool.max_bytes = total_b / 2;
    *out = c;
    return STAN_OK;
}

void stan_hip_destroy(stan_ctx *ctx) {
    if (!ctx) return;
    if (ctx->group) { stan_group_destroy(ctx); return; }
    hipSetDevice(ctx->device);
    if (ctx->p2p && ctx->p2p->ipc) stan_p2p_ipc_release(ctx);   // (a group's resources belong to the group)
    {
        void *comm = nullptr;
        { std::lock_guard<std::mutex> lk(ctx->comm_mu); comm = ctx->comm; ctx->comm = nullptr; }
        if (comm && ctx->nccl.CommDestroy) ctx->nccl.CommDestroy(comm);
    }
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    if (ctx->side) hipStreamDestroy(ctx->side);
    if (ctx->ev_a) hipEventDestroy(ctx->ev_a);
    if (ctx->ev_b) hipEventDestroy(ctx->ev_b);
    // matrices that outlive their context keep working memory-wise: they are detached and their
    // buffers go straight back to the driver when they are freed
    for (stan_matrix *K : ctx->matrices) K->ctx = nullptr;
    stan_cg_workspace_free(ctx);
    ctx->pool.flush();
    if (ctx->h_status) hipHostFree(ctx->h_status);
    if (ctx->d_status) hipFree(ctx->d_status);
    delete ctx;
}

const char *stan_hip_last_error(stan_ctx *ctx) {
    if (ctx && ctx->group) return stan_group_last_error(ctx);
    return ctx ? ctx->err.c_str() : g_err.c_str();
}
int64_t stan_hip_last_bad_element(stan_ctx *ctx) { return ctx ? ctx->bad_elem : -1; }

int stan_hip_set_stream(stan_ctx *ctx, void *hip_stream) {
    if (!ctx) return STAN_E_ARG;
    STAN_NO_GROUP(ctx, "set_stream");
    hipSetDevice(ctx->device);
    // parked blocks of the pool are reused in stream order: drain the old stream before switching
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    ctx->own_stream = false;
    ctx->stream = (hipStream_t)hip_stream;
    if (!hip_stream) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    return STAN_OK;
}

int stan_hip_set_option(stan_ctx *ctx, int32_t option, int64_t value) {
    if (!ctx) return STAN_E_ARG;
    if (ctx->group && option == STAN_OPT_COMM_P2P) return stan_group_set_p2p(ctx, value != 0);
    if (ctx->group)
        return stan_group_ctx_call(ctx, [&](stan_ctx *c) { return stan_hip_set_option(c, option, value); });
    if (option == STAN_OPT_COMM_P2P) {
        // one process per GPU: the ranks map each other's mailboxes, counters and vectors through HIP IPC; the
        // handles travel over the communicator, so EVERY rank must make this call (it is a collective)
        if (value != 0) {
            if (ctx->nranks < 2 || !ctx->comm) { ctx->err = "STAN_OPT_COMM_P2P: needs a communicator of several ranks (stan_hip_comm_init) or a handle of stan_hip_init_multi"; return STAN_E_UNSUPPORTED; }
            HIPCHK(ctx, hipSetDevice(ctx->device));
            STANCHK(stan_p2p_ipc_setup(ctx));
        }
        ctx->comm_p2p = value != 0 && ctx->p2p != nullptr;
        return STAN_OK;
    }
    if (option == STAN_OPT_CG_MERIT_STOP) ctx->cg_merit_stop = value != 0;
    else if (option == STAN_OPT_CG_RUPDATE && value >= 0 && value < (1 << 30)) ctx->cg_rupdate = (int)value;
    else if (option == STAN_OPT_ASSEMBLY_MODE && (value == 0 || value == 1)) ctx->assembly_mode = (int)value;
    else if (option == STAN_OPT_CG_FUSED_REFRESH) ctx->cg_fused_refresh = value != 0;
    else if (option == STAN_OPT_CG_SINGLE_REDUCE) ctx->cg_single_reduce = value != 0;
    else if (option == STAN_OPT_CG_FOLD_REDUCE) ctx->cg_fold_reduce = value != 0;
    else if (option == STAN_OPT_CG_REFINE && value >= 0 && value <= 2) ctx->cg_refine = (int)value;
    else if (option == STAN_OPT_CG_LAZY_SCALING) ctx->cg_lazy_scaling = value != 0;
    else if (option == STAN_OPT_VEC_STORE_NT && value >= 0 && value <= 3) ctx->vec_store_nt = (int)value;
    else if (option == STAN_OPT_PACKED_COLUMNS) ctx->cols16 = value != 0;
    else if (option == STAN_OPT_CG_DEFER_X) ctx->cg_defer_x = value != 0;
    else if (option == STAN_OPT_SPMV_SMALL && value >= 0) ctx->spmv_small_rows = value == 1 ? 150000 : value;
    else if (option == STAN_OPT_POOL) {
        ctx->pool.enabled = value != 0;
        if (!ctx->pool.enabled) {
            hipSetDevice(ctx->device);
            hipStreamSynchronize(ctx->stream);
            ctx->pool.flush();
        }
    }
    else if (option == STAN_OPT_POOL_MAX_BYTES && value >= -1) {
        hipSetDevice(ctx->device);
        if (value == -1) {   // "this process owns the device": nine tenths of what is free now
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); ctx->err = "set_option: hipMemGetInfo failed"; return STAN_E_HIP; }
            value = (int64_t)(free_b - free_b / 10 + ctx->pool.bytes_avail);   // (parked blocks are not "free" to the driver)
        }
        ctx->pool.max_bytes = (size_t)value;
        hipStreamSynchronize(ctx->stream);  // parked blocks may still be in use by queued work
        while (!ctx->pool.avail.empty() && ctx->pool.bytes_avail > ctx->pool.max_bytes) {
            hipFree(ctx->pool.avail.front().p);
            ctx->pool.bytes_avail -= ctx->pool.avail.front().cap;
            ctx->pool.avail.erase(ctx->pool.avail.begin());
        }
    }
    else if (option == STAN_OPT_OVERLAP_HALO) ctx->overlap_halo = value != 0;
    else if (option == STAN_OPT_SELL_SIGMA && value >= 1 && value <= 32) ctx->sell_sigma = (int)value;
    else if (option == STAN_OPT_PLACEMENT_TRIES && value >= 1 && value <= 64) ctx->placement_tries = (int)value;
    else if (option == STAN_OPT_PLACEMENT_MAX_BYTES && value >= 0) ctx->placement_max_bytes = value;
    else if (option == STAN_OPT_ROW_FOLDING && value >= -1 && value <= 1) ctx->row_folding = (int)value;
    else if (option == STAN_OPT_SPMV_VARIANT && (value == -1 || value == 0 || value == 9 || value == 12 || value == 20))
        ctx->spmv_variant = (int)value;
    else { ctx->err = "set_option: unknown option or bad value"; return STAN_E_ARG; }
    return STAN_OK;
}

int stan_hip_pool_info(stan_ctx *ctx, int64_t *bytes_parked, int64_t *blocks_parked) {
    if (!ctx) return STAN_E_ARG;
    if (ctx->group) ctx = stan_group_rank0(ctx);
    if (bytes_parked) *bytes_parked = (int64_t)ctx->pool.bytes_avail;
    if (blocks_parked) *blocks_parked = (int64_t)ctx->pool.avail.size();
    return STAN_OK;
}

// which transport the sharded CG of this context will use (bench.py prints it next to a multi-GPU line)
int stan_hip_comm_info(stan_ctx *ctx, int32_t *rccl_version, int32_t *comm_ranks, int32_t *comm_rank, int32_t *p2p) {
    if (!ctx) return STAN_E_ARG;
    stan_ctx *c = ctx->group ? stan_group_rank0(ctx) : ctx;
    int v = 0, n = 0, r = 0;
    stan_comm_info(c, &v, &n, &r);
    if (rccl_version) *rccl_version = v;
    if (comm_ranks) *comm_ranks = n;
    if (comm_rank) *comm_rank = r;
    if (p2p) *p2p = (c->p2p && c->comm_p2p) ? 1 : 0;
    return STAN_OK;
}

// which FILE the RCCL entry points of this context were resolved from (empty before comm_init)
int stan_hip_comm_library(stan_ctx *ctx, char *path, int64_t capacity, int32_t *reused) {
    if (!ctx || (capacity > 0 && !path) || capacity < 0) return STAN_E_ARG;
    stan_ctx *c = ctx->group ? stan_group_rank0(ctx) : ctx;
    std::string p;
    int r = 0;
    stan_comm_library(c, &p, &r);
    if (capacity > 0) {
        const size_t n = p.size() < (size_t)capacity - 1 ? p.size() : (size_t)capacity - 1;
        memcpy(path, p.data(), n);
        path[n] = 0;
    }
    if (reused) *reused = r;
    return STAN_OK;
}

int stan_hip_set_profiling(stan_ctx *ctx, int32_t enabled) {
    if (!ctx) return STAN_E_ARG;
    if (ctx->group)
        return stan_group_ctx_call(ctx, [&](stan_ctx *c) { return stan_hip_set_profiling(c, enabled); });
    ctx->profiling = enabled != 0;
    return STAN_OK;
}
int stan_hip_get_profile_rank(stan_ctx *ctx, int32_t rank, stan_profile *out) {
    if (!ctx || !out || rank < 0) return STAN_E_ARG;
    if (ctx->group) {
        if (rank >= stan_group_size(ctx)) return STAN_E_ARG;
        return stan_hip_get_profile(stan_group_rank(ctx, rank), out);
    }
    if (rank != 0) return STAN_E_ARG;
    return stan_hip_get_profile(ctx, out);
}
int stan_hip_device_info(stan_ctx *ctx, int32_t rank, int32_t *hip_ordinal, char bus_id[32]) {
    if (!ctx || rank < 0) return STAN_E_ARG;
    stan_ctx *c = ctx;
    if (ctx->group) {
        if (rank >= stan_group_size(ctx)) return STAN_E_ARG;
        c = stan_group_rank(ctx, rank);
    } else if (rank != 0) return STAN_E_ARG;
    if (hip_ordinal) *hip_ordinal = c->device;
    if (bus_id) {
        bus_id[0] = 0;
        if (hipDeviceGetPCIBusId(bus_id, 32, c->device) != hipSuccess) { (void)hipGetLastError(); bus_id[0] = 0; }
    }
    return STAN_OK;
}
int stan_hip_get_profile(stan_ctx *ctx, stan_profile *out) {
    if (!ctx || !out) return STAN_E_ARG;
    if (ctx->group) ctx = stan_group_rank0(ctx);   // rank 0's timings; spmv_bytes is that shard's
    *out = ctx->prof;
    out->assembly_colours = ctx->prof_colours;
    out->placement_candidates = ctx->prof_placement_candidates;
    out->placement_ms_best = ctx->prof_placement_ms_best;
    out->placement_ms_worst = ctx->prof_placement_ms_worst;
    out->placement_moved_vectors = ctx->prof_placement_moved_vectors;
    return STAN_OK;
}

int stan_hip_assemble_hex8_dev(stan_ctx *ctx, int64_t n_nodes, const double *d_xyz,
                               const int32_t *d_node_dof, int64_t n_elem, const int32_t *d_conn,
                               const int32_t *d_elem_mat, const uint8_t *d_elem_type,
                               int32_t n_mat, const double *mat_E_nu, int64_t n_dof,
                               const int32_t *d_red, stan_matrix **outK) {
    if (!ctx || !outK || !d_xyz || !d_node_dof || !d_red || !mat_E_nu ||
        (n_elem > 0 && (!d_conn || !d_elem_mat || !d_elem_type)))
        return STAN_E_ARG;
    STAN_NO_GROUP(ctx, "assemble_hex8_dev (device pointers belong to one device)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return stan_assemble_device(ctx, n_nodes, d_xyz, d_node_dof, n_elem, d_conn, d_elem_mat,
                                d_elem_type, n_mat, mat_E_nu, n_dof, d_red, outK);
}

int stan_hip_assemble_hex8(stan_ctx *ctx, int64_t n_nodes, const double *xyz,
                           const int32_t *node_dof, int64_t n_elem, const int32_t *conn,
                           const int32_t *elem_mat, const uint8_t *elem_type, int32_t n_mat,
                           const double *mat_E_nu, int64_t n_dof, const int32_t *red,
                           stan_matrix **outK) {
    if (!ctx || !outK || !xyz || !node_dof || !red || !mat_E_nu || n_nodes <= 0 || n_dof <= 0 ||
        n_elem < 0 || (n_elem > 0 && (!conn || !elem_mat || !elem_type))) {
        if (ctx) ctx->err = "assemble_hex8: null or empty argument";
        return STAN_E_ARG;
    }
    if (ctx->group)
        return stan_group_assemble(ctx, n_nodes, xyz, node_dof, n_elem, conn, elem_mat, elem_type, n_mat,
                                   mat_E_nu, n_dof, red, outK);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (int64_t e = 0; e < n_elem; e++) {
        if (elem_mat[e] < 0 || elem_mat[e] >= n_mat) {
            ctx->err = "assemble_hex8: elem_mat out of range at element " + std::to_string(e);
            return STAN_E_ARG;
        }
        if (elem_type[e] != STAN_HEX8_G1 && elem_type[e] != STAN_HEX8_G2) {
            ctx->err = "assemble_hex8: unsupported element type at element " + std::to_string(e);
            return STAN_E_UNSUPPORTED;
        }
    }
    // A rank of a sharded run only needs the elements that touch its block rows (boundary elements
    // are assembled by both owners): they are picked on the host, in element order (the
    // accumulation order of a row is the element order, so K keeps its bits), and only they go to
    // the device -- neither the upload nor the incidence scan of a rank grows with the ranks of
    // the others.  The node arrays stay whole: any node may be a halo column.
    std::vector<int32_t> sub, conn_s, mat_s;
    std::vector<uint8_t> type_s;
    const bool shard = ctx->nranks > 1 && n_elem > 0;
    if (shard) {
        const int64_t nb = n_dof / 3;
        const int64_t r0 = stan_row_start(nb, ctx->nranks, ctx->rank), r1 = stan_row_start(nb, ctx->nranks, ctx->rank + 1);
        for (int64_t e = 0; e < n_elem; e++) {
            bool mine = false;
            for (int a = 0; a < 8; a++) {
                const int32_t nd = conn[e * 8 + a];
                if (nd < 0 || nd >= n_nodes) {
                    ctx->err = "assemble: connectivity references a node index outside [0,n_nodes)";
                    return STAN_E_ARG;
                }
                const int64_t row = node_dof[3 * (int64_t)nd] / 3;   // a bad DOF layout is caught on the device
                mine |= row >= r0 && row < r1;
            }
            if (mine) sub.push_back((int32_t)e);
        }
        conn_s.resize(sub.size() * 8); mat_s.resize(sub.size()); type_s.resize(sub.size());
        for (size_t i = 0; i < sub.size(); i++) {
            memcpy(&conn_s[8 * i], conn + 8 * (int64_t)sub[i], 8 * sizeof(int32_t));
            mat_s[i] = elem_mat[sub[i]];
            type_s[i] = elem_type[sub[i]];
        }
        conn = conn_s.data(); elem_mat = mat_s.data(); elem_type = type_s.data();
        n_elem = (int64_t)sub.size();
    }
    dbuf<double> dx; dbuf<int32_t> dd, dc, dm, dr; dbuf<uint8_t> dt;
    STANCHK(dx.upload(ctx, xyz, (size_t)n_nodes * 3));
    STANCHK(dd.upload(ctx, node_dof, (size_t)n_nodes * 3));
    STANCHK(dc.upload(ctx, conn, (size_t)n_elem * 8));
    STANCHK(dm.upload(ctx, elem_mat, (size_t)n_elem));
    STANCHK(dt.upload(ctx, elem_type, (size_t)n_elem));
    STANCHK(dr.upload(ctx, red, (size_t)n_dof));
    const int rc = stan_assemble_device(ctx, n_nodes, dx.p, dd.p, n_elem, dc.p, dm.p, dt.p, n_mat, mat_E_nu,
                                        n_dof, dr.p, outK);
    if (rc == STAN_OK) (*outK)->n_elem_scanned = n_elem;
    if (rc == STAN_E_DETJ && shard && ctx->bad_elem >= 0 && ctx->bad_elem < (int64_t)sub.size()) {
        ctx->bad_elem = sub[(size_t)ctx->bad_elem];   // element number of the whole model
        ctx->err = "det J == 0 in element " + std::to_string(ctx->bad_elem) + " (MatrixST.Inverse would throw)";
    }
    // the uploads above are stream-ordered with the assembly; the host vectors must outlive them
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return rc;
}

void stan_hip_matrix_free(stan_matrix *K) {
    if (!K) return;
    if (!K->parts.empty()) { stan_group_matrix_free(K); return; }
    if (K->ctx) {
        hipSetDevice(K->ctx->device);
        auto &v = K->ctx->matrices;
        for (size_t i = 0; i < v.size(); i++)
            if (v[i] == K) { v.erase(v.begin() + i); break; }
    }
    // the solves that used these buffers have been synchronised by their own calls; the blocks go
    // back to the context's pool (stan_pool) or to the driver
    for (void *q : {(void *)K->d_slot_ptr, (void *)K->d_rowlen, (void *)K->d_rowof, (void *)K->d_posof, (void *)K->d_cols, (void *)K->d_vals,
                    (void *)K->d_vals32, (void *)K->d_vals48, (void *)K->d_cols16, (void *)K->d_colbase,
                    (void *)K->d_pair_ptr, (void *)K->d_slice_packed, (void *)K->d_red, (void *)K->d_fixmask,
                    (void *)K->d_scale, (void *)K->d_send_rows, (void *)K->d_halo_glob, (void *)K->d_sendbuf,
                    (void *)K->d_sl_int, (void *)K->d_sl_bnd, (void *)K->d_fold_ptr, (void *)K->d_fold_meta,
                    (void *)K->d_fold_plan, (void *)K->d_fold_cols, (void *)K->d_fold_cols16, (void *)K->d_fold_colbase,
                    (void *)K->d_fold_pair_ptr, (void *)K->d_fold_packed, (void *)K->d_fold_vals, (void *)K->d_fold_vals32, (void *)K->d_fold_vals48})
        stan_dfree(K->ctx, q);
    delete K;
}

int stan_hip_cg_solve_dev(stan_ctx *ctx, stan_matrix *K, const double *d_F, double eps_f,
                          int32_t max_its, int32_t precision_mode, double *d_U,
                          int32_t *termination_type, int32_t *iterations, double *rel_residual) {
    if (!ctx || !K || !d_F || !d_U || K->ctx != ctx) return STAN_E_ARG;
    STAN_NO_GROUP(ctx, "cg_solve_dev (device pointers belong to one device)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return stan_cg_device(ctx, K, d_F, eps_f, max_its, precision_mode, d_U, termination_type,
                          iterations, rel_residual);
}

int stan_hip_cg_solve(stan_ctx *ctx, stan_matrix *K, const double *F, double eps_f,
                      int32_t max_its, int32_t precision_mode, double *U,
                      int32_t *termination_type, int32_t *iterations, double *rel_residual) {
    if (!ctx || !K || !F || !U || K->ctx != ctx) return STAN_E_ARG;
    if (ctx->group)
        return stan_group_cg_solve(ctx, K, F, eps_f, max_its, precision_mode, U, termination_type, iterations,
                                   rel_residual);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)K->n_red;
    dbuf<double> dF, dU;
    if (ctx->result_segment && ctx->nranks > 1) {
        // a rank of a one-process group: only ITS entries [u0, u1) of F go up and of U come down -- the ranks
        // share the caller's buffers (disjoint ranges), nothing is gathered on the devices (multi.hip)
        const size_t u0 = (size_t)K->u0, nu = (size_t)(K->u1 - K->u0);
        STANCHK(dF.alloc(ctx, N ? N : 1));
        STANCHK(dU.alloc(ctx, N ? N : 1));
        if (nu) HIPCHK(ctx, hipMemcpyAsync(dF.p + u0, F + u0, nu * 8, hipMemcpyHostToDevice, ctx->stream));
        STANCHK(stan_cg_device(ctx, K, dF.p, eps_f, max_its, precision_mode, dU.p, termination_type,
                               iterations, rel_residual));
        if (nu) HIPCHK(ctx, hipMemcpyAsync(U + u0, dU.p + u0, nu * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return STAN_OK;
    }
    STANCHK(dF.upload(ctx, F, N));
    STANCHK(dU.alloc(ctx, N));
    HIPCHK(ctx, hipMemsetAsync(dU.p, 0, (N ? N : 1) * 8, ctx->stream));
    STANCHK(stan_cg_device(ctx, K, dF.p, eps_f, max_its, precision_mode, dU.p, termination_type,
                           iterations, rel_residual));
    if (N) HIPCHK(ctx, hipMemcpyAsync(U, dU.p, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return STAN_OK;
}

int stan_hip_recover_hex8_dev(stan_ctx *ctx, int64_t n_nodes, const double *d_xyz,
                              const double *d_disp, int64_t n_elem, const int32_t *d_conn,
                              const int32_t *d_elem_mat, const uint8_t *d_elem_type,
                              int32_t n_mat, const double *mat_E_nu, double *d_strain,
                              double *d_stress) {
    if (!ctx || !d_xyz || !d_disp || !mat_E_nu || n_mat <= 0 ||
        (n_elem > 0 && (!d_conn || !d_elem_mat || !d_elem_type || !d_strain || !d_stress)))
        return STAN_E_ARG;
    STAN_NO_GROUP(ctx, "recover_hex8_dev (device pointers belong to one device)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return stan_recover_device(ctx, n_nodes, d_xyz, d_disp, n_elem, d_conn, d_elem_mat, d_elem_type,
                               n_mat, mat_E_nu, d_strain, d_stress, nullptr, nullptr, nullptr);
}

int stan_hip_recover_hex8(stan_ctx *ctx, int64_t n_nodes, const double *xyz, const double *disp,
                          int64_t n_elem, const int32_t *conn, const int32_t *elem_mat,
                          const uint8_t *elem_type, int32_t n_mat, const double *mat_E_nu,
                          double *strain, double *stress) {
    if (!ctx || !xyz || !disp || !mat_E_nu || n_nodes <= 0 || n_mat <= 0 || n_elem < 0 ||
        (n_elem > 0 && (!conn || !elem_mat || !elem_type || !strain || !stress)))
        return STAN_E_ARG;
    if (ctx->group)
        return stan_group_recover(ctx, n_nodes, xyz, disp, n_elem, conn, elem_mat, elem_type, n_mat, mat_E_nu,
                                  strain, stress);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (int64_t e = 0; e < n_elem; e++) {
        if (elem_mat[e] < 0 || elem_mat[e] >= n_mat) { ctx->err = "recover_hex8: elem_mat out of range"; return STAN_E_ARG; }
        for (int a = 0; a < 8; a++)
            if (conn[e * 8 + a] < 0 || conn[e * 8 + a] >= n_nodes) { ctx->err = "recover_hex8: node index out of range"; return STAN_E_ARG; }
    }
    dbuf<double> dx, du, de, ds; dbuf<int32_t> dc, dm; dbuf<uint8_t> dt;
    STANCHK(dx.upload(ctx, xyz, (size_t)n_nodes * 3));
    STANCHK(du.upload(ctx, disp, (size_t)n_nodes * 3));
    STANCHK(dc.upload(ctx, conn, (size_t)n_elem * 8));
    STANCHK(dm.upload(ctx, elem_mat, (size_t)n_elem));
    STANCHK(dt.upload(ctx, elem_type, (size_t)n_elem));
    STANCHK(de.alloc(ctx, (size_t)n_elem * 48));
    STANCHK(ds.alloc(ctx, (size_t)n_elem * 48));
    STANCHK(stan_recover_device(ctx, n_nodes, dx.p, du.p, n_elem, dc.p, dm.p, dt.p, n_mat, mat_E_nu,
                                de.p, ds.p, nullptr, nullptr, nullptr));
    if (n_elem) {
        HIPCHK(ctx, hipMemcpyAsync(strain, de.p, (size_t)n_elem * 48 * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(stress, ds.p, (size_t)n_elem * 48 * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return STAN_OK;
}

// ---- results kept on the device(s) (stan_hip_recover_hex8_keep) ------------------------------------------------------
}  // extern "C"
struct stan_results {
    struct part { int device; int64_t e0, e1; double *d_strain, *d_stress; };
    std::vector<part> parts;
    int64_t n_elem = 0;
};
namespace {
// pinned staging of one host thread (stan_hip_results_map): grown on demand, released when the thread ends
struct map_stage {
    double *p = nullptr;
    size_t cap = 0;   // doubles
    ~map_stage() { if (p) hipHostFree(p); }
};
thread_local map_stage g_stage;
int recover_keep_one(stan_ctx *ctx, int64_t n_nodes, const double *xyz, const double *disp, int64_t n_elem, const int32_t *conn,
                     const int32_t *elem_mat, const uint8_t *elem_type, int32_t n_mat, const double *mat_E_nu, int64_t e_base,
                     stan_results::part *out) {
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (int64_t e = 0; e < n_elem; e++) {
        if (elem_mat[e] < 0 || elem_mat[e] >= n_mat) { ctx->err = "recover_hex8: elem_mat out of range"; return STAN_E_ARG; }
        for (int a = 0; a < 8; a++)
            if (conn[e * 8 + a] < 0 || conn[e * 8 + a] >= n_nodes) { ctx->err = "recover_hex8: node index out of range"; return STAN_E_ARG; }
    }
    dbuf<double> dx, du; dbuf<int32_t> dc, dm; dbuf<uint8_t> dt;
    STANCHK(dx.upload(ctx, xyz, (size_t)n_nodes * 3));
    STANCHK(du.upload(ctx, disp, (size_t)n_nodes * 3));
    STANCHK(dc.upload(ctx, conn, (size_t)n_elem * 8));
    STANCHK(dm.upload(ctx, elem_mat, (size_t)n_elem));
    STANCHK(dt.upload(ctx, elem_type, (size_t)n_elem));
    // plain device memory, owned by the results object (it may outlive the context's pool)
    double *de = nullptr, *ds = nullptr;
    const size_t bytes = (size_t)(n_elem > 0 ? n_elem : 1) * 48 * 8;
    auto both = [&]() {
        if (hipMalloc((void **)&de, bytes) == hipSuccess && hipMalloc((void **)&ds, bytes) == hipSuccess) return true;
        (void)hipGetLastError();
        if (de) hipFree(de);
        de = ds = nullptr;
        return false;
    };
    if (!both()) {
        // the context's pool may hold most of the device parked (stan_solver lets it: STAN_OPT_POOL_MAX_BYTES = -1) while K
        // and the CG's vectors are still live: give the parked blocks back -- what stan_dmalloc_bytes does for a pooled
        // request -- and try once more (400^3 keeps 49 GB of results)
        if (!ctx->pool.avail.empty() && hipStreamSynchronize(ctx->stream) == hipSuccess) {
            ctx->pool.flush();
            (void)both();
        }
        if (!de) {
            ctx->err = "recover_hex8_keep: device allocation of 2 x " + std::to_string(bytes) + " B failed";
            return STAN_E_ALLOC;
        }
    }
    int rc = stan_recover_device(ctx, n_nodes, dx.p, du.p, n_elem, dc.p, dm.p, dt.p, n_mat, mat_E_nu, de, ds, nullptr, nullptr, nullptr);
    if (rc == STAN_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = STAN_E_HIP;
    if (rc != STAN_OK) { hipFree(de); hipFree(ds); return rc; }
    *out = stan_results::part{ctx->device, e_base, e_base + n_elem, de, ds};
    return STAN_OK;
}
}  // namespace
extern "C" {

int stan_hip_recover_hex8_keep(stan_ctx *ctx, int64_t n_nodes, const double *xyz, const double *disp,
                               int64_t n_elem, const int32_t *conn, const int32_t *elem_mat,
                               const uint8_t *elem_type, int32_t n_mat, const double *mat_E_nu, stan_results **out) {
    if (!ctx || !out || !xyz || !disp || !mat_E_nu || n_nodes <= 0 || n_mat <= 0 || n_elem < 0 ||
        (n_elem > 0 && (!conn || !elem_mat || !elem_type)))
        return STAN_E_ARG;
    *out = nullptr;
    stan_results *res = new stan_results();
    res->n_elem = n_elem;
    int rc = STAN_OK;
    if (ctx->group) {
        const int n = stan_group_size(ctx);
        res->parts.assign((size_t)n, stan_results::part{0, 0, 0, nullptr, nullptr});
        rc = stan_group_ctx_call_ranked(ctx, [&](stan_ctx *c, int r) {
            const int64_t e0 = n_elem * r / n, e1 = n_elem * (r + 1) / n;
            res->parts[(size_t)r] = stan_results::part{c->device, e0, e0, nullptr, nullptr};
            if (e1 <= e0) return (int)STAN_OK;
            const int e = recover_keep_one(c, n_nodes, xyz, disp, e1 - e0, conn + 8 * e0, elem_mat + e0, elem_type + e0, n_mat,
                                           mat_E_nu, e0, &res->parts[(size_t)r]);
            if (e == STAN_E_UNSUPPORTED || e == STAN_E_DETJ) {   // element numbers of the whole model
                ctx->bad_elem = c->bad_elem + e0;
                c->err = (e == STAN_E_DETJ ? "det J == 0 in element " : "stress recovery: HEX8_G1 element ") + std::to_string(ctx->bad_elem) +
                         (e == STAN_E_DETJ ? "" : " (the reference throws: N has one row, Element.cs:242)");
            }
            return e;
        });
    } else {
        res->parts.assign(1, stan_results::part{ctx->device, 0, 0, nullptr, nullptr});
        if (n_elem > 0) rc = recover_keep_one(ctx, n_nodes, xyz, disp, n_elem, conn, elem_mat, elem_type, n_mat, mat_E_nu, 0, &res->parts[0]);
    }
    if (rc != STAN_OK) { stan_hip_results_free(res); return rc; }
    *out = res;
    return STAN_OK;
}

int stan_hip_results_map(stan_results *res, int64_t e0, int64_t e1, const double **strain, const double **stress) {
    if (!res || !strain || !stress || e0 < 0 || e1 < e0 || e1 > res->n_elem) return STAN_E_ARG;
    const size_t n = (size_t)(e1 - e0) * 48;
    map_stage &st = g_stage;
    if (st.cap < 2 * n) {
        if (st.p) hipHostFree(st.p);
        st.p = nullptr; st.cap = 0;
        const size_t want = 2 * n > (size_t)1 << 16 ? 2 * n : (size_t)1 << 16;
        if (hipHostMalloc((void **)&st.p, want * 8, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); st.p = nullptr; return STAN_E_ALLOC; }
        st.cap = want;
    }
    for (const stan_results::part &pt : res->parts) {
        const int64_t a = e0 > pt.e0 ? e0 : pt.e0, b = e1 < pt.e1 ? e1 : pt.e1;
        if (b <= a) continue;
        if (hipSetDevice(pt.device) != hipSuccess) return STAN_E_HIP;
        const size_t off = (size_t)(a - e0) * 48, cnt = (size_t)(b - a) * 48 * 8;
        // blocking copies on the null stream of the calling thread: the kernel that produced the data was
        // synchronised by stan_hip_recover_hex8_keep, the library's own streams are non-blocking
        if (hipMemcpy(st.p + off, pt.d_strain + (size_t)(a - pt.e0) * 48, cnt, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(st.p + n + off, pt.d_stress + (size_t)(a - pt.e0) * 48, cnt, hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipGetLastError();
            return STAN_E_HIP;
        }
    }
    *strain = st.p;
    *stress = st.p + n;
    return STAN_OK;
}

void stan_hip_results_free(stan_results *res) {
    if (!res) return;
    for (stan_results::part &pt : res->parts) {
        if (!pt.d_strain && !pt.d_stress) continue;
        hipSetDevice(pt.device);
        if (pt.d_strain) hipFree(pt.d_strain);
        if (pt.d_stress) hipFree(pt.d_stress);
    }
    delete res;
}

int stan_hip_nodal_forces_hex8(stan_ctx *ctx, int64_t n_nodes, const double *xyz, const double *disp,
                               const int32_t *node_dof, int64_t n_elem, const int32_t *conn,
                               const int32_t *elem_mat, const uint8_t *elem_type, int32_t n_mat,
                               const double *mat_E_nu, int64_t n_dof, double *elem_forces, double *R) {
    if (!ctx || !xyz || !disp || !node_dof || !mat_E_nu || n_nodes <= 0 || n_mat <= 0 || n_elem < 0 ||
        n_dof != n_nodes * 3 || (!elem_forces && !R) || (n_elem > 0 && (!conn || !elem_mat || !elem_type)))
        return STAN_E_ARG;
    if (ctx->group)   // R is an all-element sum: one device (rank 0); its error text follows the handle
        return stan_group_rank0_call(ctx, [&](stan_ctx *c) {
            return stan_hip_nodal_forces_hex8(c, n_nodes, xyz, disp, node_dof, n_elem, conn, elem_mat, elem_type, n_mat,
                                              mat_E_nu, n_dof, elem_forces, R);
        });
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (int64_t e = 0; e < n_elem; e++) {
        if (elem_mat[e] < 0 || elem_mat[e] >= n_mat) { ctx->err = "nodal_forces_hex8: elem_mat out of range"; return STAN_E_ARG; }
        for (int a = 0; a < 8; a++)
            if (conn[e * 8 + a] < 0 || conn[e * 8 + a] >= n_nodes) { ctx->err = "nodal_forces_hex8: node index out of range"; return STAN_E_ARG; }
    }
    for (int64_t k = 0; k < n_dof; k++)
        if (node_dof[k] < 0 || node_dof[k] >= n_dof) { ctx->err = "nodal_forces_hex8: DOF out of range"; return STAN_E_DOF_LAYOUT; }
    dbuf<double> dx, du, df, dR; dbuf<int32_t> dc, dm, dd; dbuf<uint8_t> dt;
    STANCHK(dx.upload(ctx, xyz, (size_t)n_nodes * 3));
    STANCHK(du.upload(ctx, disp, (size_t)n_nodes * 3));
    STANCHK(dd.upload(ctx, node_dof, (size_t)n_nodes * 3));
    STANCHK(dc.upload(ctx, conn, (size_t)n_elem * 8));
    STANCHK(dm.upload(ctx, elem_mat, (size_t)n_elem));
    STANCHK(dt.upload(ctx, elem_type, (size_t)n_elem));
    if (elem_forces) STANCHK(df.alloc(ctx, (size_t)n_elem * 24));
    if (R) {
        STANCHK(dR.alloc(ctx, (size_t)n_dof));
        HIPCHK(ctx, hipMemsetAsync(dR.p, 0, (size_t)n_dof * 8, ctx->stream));
    }
    STANCHK(stan_recover_device(ctx, n_nodes, dx.p, du.p, n_elem, dc.p, dm.p, dt.p, n_mat, mat_E_nu,
                                nullptr, nullptr, dd.p, elem_forces ? df.p : nullptr, R ? dR.p : nullptr));
    if (elem_forces && n_elem)
        HIPCHK(ctx, hipMemcpyAsync(elem_forces, df.p, (size_t)n_elem * 24 * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (R) HIPCHK(ctx, hipMemcpyAsync(R, dR.p, (size_t)n_dof * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return STAN_OK;
}

int stan_hip_matrix_part_info(stan_matrix *K, int32_t part, stan_matrix_info *o) {
    if (!K || !o || part < 0) return STAN_E_ARG;
    if (K->parts.empty()) return part == 0 ? stan_hip_matrix_info(K, o) : STAN_E_ARG;
    if ((size_t)part >= K->parts.size()) return STAN_E_ARG;
    return stan_hip_matrix_info(K->parts[(size_t)part], o);
}

int stan_hip_matrix_info(stan_matrix *K, stan_matrix_info *o) {
    if (!K || !o) return STAN_E_ARG;
    o->n_dof = K->n_dof; o->n_reduced = K->n_red; o->n_block_rows = K->nb_glob;
    o->row_begin = K->r0; o->row_end = K->r1; o->n_halo = K->nhalo; o->n_blocks = K->nblocks;
    o->n_slots = K->nslots;
    o->bytes_matrix = K->nslots * 64 * (9 * 8 + 4);
    o->scaled = (K->parts.empty() ? K->scaled : K->parts[0]->scaled) ? 1 : 0;
    o->max_row_blocks = K->max_row_blocks;
    o->n_elements_on_device = K->n_elem_scanned;
    o->sell_sigma = K->parts.empty() ? K->sigma : K->parts[0]->sigma;
    const stan_matrix *P = K->parts.empty() ? K : K->parts[0];
    o->folded_slots_permille = P->fold_state == 1 && P->nslots > 0 ? (int32_t)(1000 * P->nfslots / P->nslots) : 0;
    return STAN_OK;
}

int stan_hip_ke_hex8_batch(stan_ctx *ctx, int64_t n, const double *xyz8, double E, double nu,
                           const uint8_t *type, double *out) {
    if (!ctx || n < 0 || (n > 0 && (!xyz8 || !type || !out))) return STAN_E_ARG;
    if (ctx->group)
        return stan_group_rank0_call(ctx, [&](stan_ctx *c) { return stan_hip_ke_hex8_batch(c, n, xyz8, E, nu, type, out); });
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (int64_t e = 0; e < n; e++)
        if (type[e] != STAN_HEX8_G1 && type[e] != STAN_HEX8_G2) {
            ctx->err = "ke_hex8: unsupported element type";
            return STAN_E_UNSUPPORTED;
        }
    dbuf<double> dx, dk; dbuf<uint8_t> dt;
    STANCHK(dx.upload(ctx, xyz8, (size_t)n * 24));
    STANCHK(dt.upload(ctx, type, (size_t)n));
    STANCHK(dk.alloc(ctx, (size_t)n * 576));
    STANCHK(stan_ke_batch_device(ctx, n, dx.p, E, nu, dt.p, dk.p));
    if (n) HIPCHK(ctx, hipMemcpyAsync(out, dk.p, (size_t)n * 576 * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return STAN_OK;
}

int stan_hip_ke_hex8(stan_ctx *ctx, const double xyz8[24], double E, double nu, int32_t type,
                     double out[576]) {
    uint8_t t = (uint8_t)type;
    if (type != STAN_HEX8_G1 && type != STAN_HEX8_G2) {
        if (ctx) ctx->err = "ke_hex8: unsupported element type";
        return STAN_E_UNSUPPORTED;
    }
    return stan_hip_ke_hex8_batch(ctx, 1, xyz8, E, nu, &t, out);
}

int stan_hip_matrix_to_csr(stan_ctx *ctx, stan_matrix *K, int32_t upper_only, int64_t *nnz,
                           int64_t *rowptr, int32_t *col, double *val) {
    if (!ctx || !K || !nnz || K->ctx != ctx) return STAN_E_ARG;
    STAN_NO_GROUP(ctx, "matrix_to_csr");
    if (ctx->nranks != 1) { ctx->err = "matrix_to_csr: single-rank contexts only"; return STAN_E_UNSUPPORTED; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // export K itself, not S K S.  A matrix no solve has scaled yet is exported AS ASSEMBLED -- the exact bits, and the
    // export changes nothing (round 5 scaled it here as a side effect, which also cost the next solve its lazy-scaling
    // first product: ADVICE r05).  Once a solve has brought it into its scaled form, the values are divided by
    // s_row s_col on the way out: within 1 ulp of the assembled entry ((a t) / t), the matrix again untouched.
    const bool scaled = K->scaled;
    const int64_t nloc = K->nloc;
    std::vector<double> scale((size_t)3 * (size_t)K->nslices * 64);
    std::vector<int32_t> slot_ptr((size_t)K->nslices + 1), rowlen((size_t)K->nslices * 64),
        posof((size_t)K->nslices * 64), cols((size_t)K->nslots * 64), red((size_t)K->n_dof);
    std::vector<double> vals((size_t)K->nslots * 9 * 64);
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(slot_ptr.data(), K->d_slot_ptr, slot_ptr.size() * 4, hipMemcpyDeviceToHost, st));
    if (!rowlen.empty()) HIPCHK(ctx, hipMemcpyAsync(rowlen.data(), K->d_rowlen, rowlen.size() * 4, hipMemcpyDeviceToHost, st));
    if (!posof.empty()) HIPCHK(ctx, hipMemcpyAsync(posof.data(), K->d_posof, posof.size() * 4, hipMemcpyDeviceToHost, st));
    if (!cols.empty()) HIPCHK(ctx, hipMemcpyAsync(cols.data(), K->d_cols, cols.size() * 4, hipMemcpyDeviceToHost, st));
    if (!vals.empty()) HIPCHK(ctx, hipMemcpyAsync(vals.data(), K->d_vals, vals.size() * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(red.data(), K->d_red, red.size() * 4, hipMemcpyDeviceToHost, st));
    if (scaled && !scale.empty()) HIPCHK(ctx, hipMemcpyAsync(scale.data(), K->d_scale, scale.size() * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    int64_t count = 0;
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1 && (!rowptr || !col || !val)) break;
        int64_t q = 0;
        for (int64_t row = 0; row < nloc; row++) {
            const int64_t sl = posof[(size_t)row] >> 6;   // SELL-C-sigma: where the row sits in the sliced layout
            const int lane = (int)(posof[(size_t)row] & 63);
            for (int m = 0; m < 3; m++) {
                const int64_t d = 3 * row + m;
                if (red[d] == -1) continue;
                if (pass == 1) rowptr[d - red[d]] = q;
                for (int k = 0; k < rowlen[row]; k++) {
                    const int64_t slot = (int64_t)slot_ptr[sl] + k;
                    const int32_t c = cols[slot * 64 + lane];
                    for (int n = 0; n < 3; n++) {
                        const int64_t dc = 3 * (int64_t)c + n;
                        if (red[dc] == -1) continue;
                        if (upper_only && dc < d) continue;
                        if (pass == 1) {
                            col[q] = (int32_t)(dc - red[dc]);
                            const double a = vals[(slot * 9 + 3 * m + n) * 64 + lane];
                            val[q] = scaled ? a / (scale[(size_t)d] * scale[(size_t)dc]) : a;
                        }
                        q++;
                    }
                }
            }
        }
        if (pass == 0) count = q;
        else rowptr[K->n_red] = q;
    }
    *nnz = count;
    return STAN_OK;
}

int stan_hip_spmv(stan_ctx *ctx, stan_matrix *K, const double *x, double *y) {
    if (!ctx || !K || !x || !y || K->ctx != ctx) return STAN_E_ARG;
    STAN_NO_GROUP(ctx, "spmv");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)K->n_red;
    dbuf<double> dx, dy;
    STANCHK(dx.upload(ctx, x, N));
    STANCHK(dy.alloc(ctx, N));
    STANCHK(stan_spmv_reduced(ctx, K, dx.p, dy.p));
    if (N) HIPCHK(ctx, hipMemcpyAsync(y, dy.p, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return STAN_OK;
}

int stan_hip_matrix_diagonal(stan_ctx *ctx, stan_matrix *K, double *diag) {
    if (!ctx || !K || !diag || K->ctx != ctx) return STAN_E_ARG;
    STAN_NO_GROUP(ctx, "matrix_diagonal");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)K->n_red;
    dbuf<double> dd;
    STANCHK(dd.alloc(ctx, N ? N : 1));
    STANCHK(stan_matrix_diagonal(ctx, K, dd.p));
    if (N) HIPCHK(ctx, hipMemcpyAsync(diag, dd.p, N * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return STAN_OK;
}

int stan_hip_matrix_plan(stan_ctx *ctx, stan_matrix *K, int64_t *row_starts, int64_t *n_halo,
                         int32_t *halo_glob, int32_t *n_nbr, int32_t *nbr, int64_t *send_off,
                         int32_t *send_rows, int64_t *recv_off) {
    if (!ctx || !K || K->ctx != ctx) return STAN_E_ARG;
    STAN_NO_GROUP(ctx, "matrix_plan");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (row_starts) for (size_t i = 0; i < K->row_starts.size(); i++) row_starts[i] = K->row_starts[i];
    if (n_halo) *n_halo = K->nhalo;
    if (n_nbr) *n_nbr = (int32_t)K->nbr.size();
    if (nbr) for (size_t i = 0; i < K->nbr.size(); i++) nbr[i] = K->nbr[i];
    if (send_off) { send_off[0] = 0; for (size_t i = 0; i < K->send_off.size(); i++) send_off[i] = K->send_off[i]; }
    if (recv_off) { recv_off[0] = 0; for (size_t i = 0; i < K->recv_off.size(); i++) recv_off[i] = K->recv_off[i]; }
    if (halo_glob && K->nhalo)
        HIPCHK(ctx, hipMemcpy(halo_glob, K->d_halo_glob, (size_t)K->nhalo * 4, hipMemcpyDeviceToHost));
    if (send_rows && !K->send_off.empty() && K->send_off.back())
        HIPCHK(ctx, hipMemcpy(send_rows, K->d_send_rows, (size_t)K->send_off.back() * 4, hipMemcpyDeviceToHost));
    return STAN_OK;
}

int stan_hip_spmv_local(stan_ctx *ctx, stan_matrix *K, const double *x_local, double *y_owned) {
    if (!ctx || !K || !x_local || !y_owned || K->ctx != ctx) return STAN_E_ARG;
    STAN_NO_GROUP(ctx, "spmv_local");
    if (K->scaled) { ctx->err = "spmv_local: matrix already carries the CG scaling"; return STAN_E_UNSUPPORTED; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t npad = (int64_t)K->nslices * 64;
    const int64_t ng = 3 * (npad > K->nloc + K->nhalo ? npad : K->nloc + K->nhalo);
    dbuf<double> dx, dy;
    STANCHK(dx.alloc(ctx, (size_t)ng));
    STANCHK(dy.alloc(ctx, (size_t)(3 * npad + 3)));
    HIPCHK(ctx, hipMemsetAsync(dx.p, 0, (size_t)ng * 8, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(dx.p, x_local, (size_t)(3 * (K->nloc + K->nhalo)) * 8, hipMemcpyHostToDevice, ctx->stream));
    STANCHK(stan_spmv_local(ctx, K, dx.p, dy.p));
    if (K->nloc) HIPCHK(ctx, hipMemcpy(y_owned, dy.p, (size_t)(3 * K->nloc) * 8, hipMemcpyDeviceToHost));
    return STAN_OK;
}

int stan_hip_spmv_bench(stan_ctx *ctx, stan_matrix *K, int32_t precision_mode, int32_t reps,
                        double *avg_ms) {
    if (!ctx || !K || !avg_ms || reps <= 0 || K->ctx != ctx) return STAN_E_ARG;
    STAN_NO_GROUP(ctx, "spmv_bench");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return stan_spmv_bench_device(ctx, K, precision_mode, reps, avg_ms);
}

int stan_hip_stream_bench(stan_ctx *ctx, stan_matrix *K, int32_t reps, double *avg_ms, int64_t *bytes) {
    if (!ctx || !K || !avg_ms || !bytes || reps <= 0 || K->ctx != ctx) return STAN_E_ARG;
    STAN_NO_GROUP(ctx, "stream_bench");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return stan_stream_bench_device(ctx, K, reps, avg_ms, bytes);
}

}  // extern "C"
