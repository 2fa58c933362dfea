// comm.hip -- RCCL plumbing of the sharded CG (one process per GPU, xGMI underneath).
//
// The reference has no distributed path (a single process with TPL
// threads); this is new design.  Rows of K are sharded by contiguous block-row ranges
// in reference DOF order.  Per CG iteration the only exchanges are
//   * one halo exchange of p (grouped ncclSend/ncclRecv with the few neighbour ranks whose
//     rows couple to ours; a BFS-level ordering keeps that to rank+-1 on connected meshes),
//   * two all-reduces of 1-2 doubles (p.Ap; r.r with the merit function piggy-backed).
// The payloads are latency-bound, not bandwidth-bound, so nothing is bucketed or ringed.
//
// RCCL is resolved with dlopen at comm_init: libstan_hip.so has no link-time dependency
// on it and loads on hosts without RCCL (and next to torch's bundled librccl.so.1).
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>

#include "internal.h"

namespace {

struct nccl_uid { char internal[128]; };
constexpr int NCCL_SUM = 0, NCCL_F64 = 8;

typedef int (*fn_commInitRank)(void **, int, nccl_uid, int);

int load_rccl(stan_ctx *ctx) {
    rccl_api &n = ctx->nccl;
    if (n.handle) return STAN_OK;
    // ONE RCCL per process: a host that already has RCCL mapped -- bench.py under torch, whose
    // libtorch_hip.so brings its bundled librccl.so (SONAME librccl.so.1) -- keeps using THAT file: RTLD_NOLOAD
    // only returns what is loaded.  A fresh load happens when there is none.  STAN_RCCL_LIB: test hook only
    // (tests/fake_rccl: several ranks on ONE GPU, which RCCL refuses).
    const char *hook = getenv("STAN_RCCL_LIB");
    if (hook && *hook) n.handle = dlopen(hook, RTLD_NOW | RTLD_GLOBAL);
    else {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (int i = 0; i < 2 && !n.handle; i++) {
            n.handle = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);
            n.reused = n.handle != nullptr;
        }
        for (int i = 0; i < 3 && !n.handle; i++) n.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    }
    if (!n.handle) {
        ctx->err = std::string("dlopen(librccl): ") + dlerror();
        return STAN_E_COMM;
    }
#define SYM(field, name)                                              \
    *(void **)(&n.field) = dlsym(n.handle, name);                     \
    if (!n.field) {                                                   \
        ctx->err = std::string("dlsym(") + name + ") failed";         \
        return STAN_E_COMM;                                           \
    }
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    *(void **)(&n.CommAbort) = dlsym(n.handle, "ncclCommAbort");   // optional
    SYM(AllReduce, "ncclAllReduce");
    SYM(Broadcast, "ncclBroadcast");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
    *(void **)(&n.GetVersion) = dlsym(n.handle, "ncclGetVersion");   // optional (bench.py prints it)
    *(void **)(&n.CommCount) = dlsym(n.handle, "ncclCommCount");
    *(void **)(&n.CommUserRank) = dlsym(n.handle, "ncclCommUserRank");
#undef SYM
    Dl_info di;   // the FILE the entry points came from: bench.py prints it next to a multi-GPU line
    if (dladdr((void *)n.AllReduce, &di) && di.dli_fname) {
        char real[4096];
        n.path = realpath(di.dli_fname, real) ? real : di.dli_fname;
    }
    return STAN_OK;
}

#define NCCLCHK(ctx, call)                                                                 \
    do {                                                                                   \
        int r_ = (call);                                                                   \
        if (r_ != 0) {                                                                     \
            (ctx)->err = std::string(#call) + ": " + (ctx)->nccl.GetErrorString(r_);       \
            return STAN_E_COMM;                                                            \
        }                                                                                  \
    } while (0)

// sendbuf[3*i + c] = vec[3*rows[i] + c]
__global__ void k_pack(int64_t n, const int32_t *rows, const double *vec, double *buf) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < 3 * n; t += stride) {
        const int64_t i = t / 3;
        const int c = (int)(t - 3 * i);
        buf[t] = vec[3 * (int64_t)rows[i] + c];
    }
}

// The communicator of this rank for one RCCL call.  The group's host thread may abort it while this
// worker is between two calls (multi.hip: a peer rank failed): pointer and flag are read under the
// context's mutex, the RCCL call itself runs outside it (an abort must be able to overtake a call that
// blocks, as the test transport's do).
void *comm_for_call(stan_ctx *ctx) {
    std::lock_guard<std::mutex> lk(ctx->comm_mu);
    return ctx->comm_broken.load() ? nullptr : ctx->comm;
}

// no communicator: fine for a single rank, an error for a detached rank of several
int no_comm(stan_ctx *ctx) {
    if (ctx->comm_broken.load()) {
        ctx->err = "the communicator was aborted after a peer rank failed";
        return STAN_E_COMM;
    }
    if (ctx->nranks == 1) return STAN_OK;
    ctx->err = "this context is a detached rank (comm_init without an id): no collectives";
    return STAN_E_COMM;
}

}  // namespace

extern "C" int stan_hip_comm_unique_id(char id[128]) {
    stan_ctx tmp;
    int rc = load_rccl(&tmp);
    if (rc) return rc;
    return tmp.nccl.GetUniqueId((void *)id) == 0 ? STAN_OK : STAN_E_COMM;
}

extern "C" int stan_hip_comm_init(stan_ctx *ctx, int rank, int nranks, const char id[128]) {
    if (!ctx || nranks < 1 || rank < 0 || rank >= nranks) return STAN_E_ARG;
    STAN_NO_GROUP(ctx, "comm_init (a multi-device handle builds its own communicator)");
    if (comm_for_call(ctx)) {
        ctx->err = "comm_init: communicator already initialised";
        return STAN_E_ARG;
    }
    ctx->rank = rank;
    ctx->nranks = nranks;
    if (!id) return STAN_OK;  // detached: partition/assembly only, no collectives (tests)
    if (nranks > 64) {
        ctx->err = "comm_init: at most 64 ranks";
        return STAN_E_ARG;
    }
    STANCHK(load_rccl(ctx));
    HIPCHK(ctx, hipSetDevice(ctx->device));
    nccl_uid uid;
    memcpy(uid.internal, id, 128);
    void *comm = nullptr;
    NCCLCHK(ctx, ((fn_commInitRank)ctx->nccl.CommInitRank)(&comm, nranks, uid, rank));
    std::lock_guard<std::mutex> lk(ctx->comm_mu);
    ctx->comm = comm;
    return STAN_OK;
}

int stan_comm_allreduce_sum_f64(stan_ctx *ctx, double *d_buf, size_t count) {
    void *comm = comm_for_call(ctx);
    if (!comm) return no_comm(ctx);
    NCCLCHK(ctx, ctx->nccl.AllReduce(d_buf, d_buf, count, NCCL_F64, NCCL_SUM, comm, ctx->stream));
    return STAN_OK;
}

// RCCL's version code and what the communicator says about itself (bench.py prints them: a SCALE line
// should say which transport it measured); zeros where there is no communicator / no such entry point
int stan_comm_info(stan_ctx *ctx, int *version, int *count, int *rank) {
    *version = *count = *rank = 0;
    void *comm = comm_for_call(ctx);
    if (ctx->nccl.GetVersion) ctx->nccl.GetVersion(version);
    if (comm && ctx->nccl.CommCount) ctx->nccl.CommCount(comm, count);
    if (comm && ctx->nccl.CommUserRank) ctx->nccl.CommUserRank(comm, rank);
    return STAN_OK;
}

// the file the RCCL entry points were resolved from, and whether it was already mapped when this library asked
int stan_comm_library(stan_ctx *ctx, std::string *path, int *reused) {
    *path = ctx->nccl.path;
    *reused = ctx->nccl.reused ? 1 : 0;
    return STAN_OK;
}

int stan_comm_halo_exchange(stan_ctx *ctx, stan_matrix *K, double *d_vec) {
    if (ctx->comm_p2p && ctx->p2p) return stan_p2p_halo_exchange(ctx, K, d_vec);   // peer to peer: no RCCL call
    void *comm = comm_for_call(ctx);
    if (!comm) return no_comm(ctx);
    if (K->nbr.empty()) return STAN_OK;
    const int64_t stot = K->send_off.back();
    if (stot > 0) {
        int64_t b = (3 * stot + 255) / 256;
        if (b > 1024) b = 1024;
        hipLaunchKernelGGL(k_pack, dim3((unsigned)b), dim3(256), 0, ctx->stream, stot, K->d_send_rows,
                           d_vec, K->d_sendbuf);
    }
    double *halo = d_vec + 3 * K->nloc;
    NCCLCHK(ctx, ctx->nccl.GroupStart());
    for (size_t i = 0; i < K->nbr.size(); i++) {
        const int64_t ns = K->send_off[i + 1] - K->send_off[i];
        const int64_t nr = K->recv_off[i + 1] - K->recv_off[i];
        if (ns > 0)
            NCCLCHK(ctx, ctx->nccl.Send(K->d_sendbuf + 3 * K->send_off[i], (size_t)(3 * ns), NCCL_F64,
                                        K->nbr[i], comm, ctx->stream));
        if (nr > 0)
            NCCLCHK(ctx, ctx->nccl.Recv(halo + 3 * K->recv_off[i], (size_t)(3 * nr), NCCL_F64,
                                        K->nbr[i], comm, ctx->stream));
    }
    NCCLCHK(ctx, ctx->nccl.GroupEnd());
    return STAN_OK;
}

// every rank contributes its owned rows of a global block vector (3 doubles per block row)
int stan_comm_allgather_rows(stan_ctx *ctx, stan_matrix *K, double *d_full) {
    void *comm = comm_for_call(ctx);
    if (!comm) return no_comm(ctx);
    NCCLCHK(ctx, ctx->nccl.GroupStart());
    for (int r = 0; r < ctx->nranks; r++) {
        const int64_t a = K->row_starts[r], b = K->row_starts[r + 1];
        if (b > a)
            NCCLCHK(ctx, ctx->nccl.Broadcast(d_full + 3 * a, d_full + 3 * a, (size_t)(3 * (b - a)),
                                             NCCL_F64, r, comm, ctx->stream));
    }
    NCCLCHK(ctx, ctx->nccl.GroupEnd());
    return STAN_OK;
}

// small host-side all-gather over the communicator (control plane of the peer-to-peer set-up: IPC handles):
// every rank contributes `bytes` bytes, all [nranks * bytes] come back; synchronises the ranks
int stan_comm_allgather_bytes(stan_ctx *ctx, const void *mine, size_t bytes, void *all) {
    void *comm = comm_for_call(ctx);
    if (!comm) return no_comm(ctx);
    unsigned char *d = nullptr;
    const size_t tot = bytes * (size_t)ctx->nranks;
    HIPCHK(ctx, hipMalloc((void **)&d, tot));
    struct F { void *p; ~F() { hipFree(p); } } fr{d};
    HIPCHK(ctx, hipMemcpyAsync(d + bytes * (size_t)ctx->rank, mine, bytes, hipMemcpyHostToDevice, ctx->stream));
    NCCLCHK(ctx, ctx->nccl.GroupStart());
    for (int r = 0; r < ctx->nranks; r++)
        NCCLCHK(ctx, ctx->nccl.Broadcast(d + bytes * (size_t)r, d + bytes * (size_t)r, bytes, 1 /* ncclUint8 */, r, comm, ctx->stream));
    NCCLCHK(ctx, ctx->nccl.GroupEnd());
    HIPCHK(ctx, hipMemcpyAsync(all, d, tot, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return STAN_OK;
}

// A peer rank failed while this one may be blocked in a collective: abort the communicator so that
// its queued work returns (ncclCommAbort), and refuse every later collective on this context.
// Called from the group's HOST thread while the rank's worker may be inside the solve: the pointer is
// taken away under the mutex (the worker's next call sees comm_broken and returns STAN_E_COMM), the
// abort itself runs outside it -- ncclCommAbort is the one RCCL call that may overtake a blocked one.
void stan_comm_abort(stan_ctx *ctx) {
    void *comm = nullptr;
    {
        std::lock_guard<std::mutex> lk(ctx->comm_mu);
        comm = ctx->comm;
        ctx->comm = nullptr;
        ctx->comm_broken.store(true);
    }
    if (comm && ctx->nccl.CommAbort) ctx->nccl.CommAbort(comm);
}
