// cg.hip -- Conjugate Gradient on gfx950: replaces SolverFunctions.LinearSolver_CG
// (SolverFunctions.cs:270-330), i.e. alglib.lincgcreate / lincgsetcond /
// lincgsolvesparse / lincgresults of alglib.net 3.16.0 (not vendored in the
// reference; its published algorithm is restated here):
//   * diagonal preconditioner applied as a symmetric scaling  A^ = S K S,
//     s_i = 1/sqrt(K_ii) (1 when K_ii <= 0), b^ = S b, result U = S x^;
//   * x0 = 0; stop when ||r^|| <= EpsF ||b^|| (type 1), after MaxIts > 0 iterations
//     (type 5), when the merit function x'Ax - 2b'x stops decreasing (type 7, previous
//     point returned), p'Ap <= 0 (type -5) or non-finite numbers (type -4);
//   * every 10th iteration the residual is recomputed as b^ - A^ x^ (extra SpMV).
//
// All kernels here are HBM-bound streaming kernels.  The scaling is folded into the
// matrix once per matrix, so an iteration is
//   SpMV (+ fused p.Ap)                      reads the matrix once
//   step  : x' = x + a p, r -= a v, r.r, merit   (one pass, 5 reads 2 writes)
//   update: p = r + b p                          (one pass, 2 reads 1 write)
// plus two 1-block reductions of per-block partial sums (fixed order => the whole solve is
// bit-reproducible).  alpha, beta and every stopping decision live in device memory; the
// host only enqueues iterations and polls a status word every CHUNK iterations, so
// there is no host synchronisation inside an iteration.
#include <chrono>
#include <cmath>
#include <thread>

#include "internal.h"
#include "p2p_device.h"

// Cache policy of the vector traffic (measured in round 2, tools/fold_ab.py, profiles/r02/fold_ab_incg_*.txt; the
// compile-time switches behind those runs are the lab build's, lab/lab_hooks.patch): the in-CG penalty of the SpMV is the
// REWRITING of its gather vector between two products, nothing else (a k_step pass in between costs
// nothing).  Rewritten by plain stores the following product ran 1.5 / 1.6 / 2.7 / 3.9 / 11 %
// slower than back to back on five boxes; by non-temporal stores (no load of the line before)
// 1.0 / 1.1 / 1.1 / 1.0 %; non-temporal stores followed by one streaming read of the vector: 0 %
// (but that read costs what it saves).  Agent- / system-scope (write-through) stores: like plain.
// k_update is a read-modify-write of p: a non-temporal store to a line its own plain load has just
// brought into L2 changes nothing, non-temporal loads AND stores recover a part (0.2 % where the
// penalty is 1.6 %).  That is what STAN_OPT_VEC_STORE_NT selects; it never costs anything.
// Writing the new p into the OTHER of two buffers (no line of it in any cache) was tried as well:
// +1.2 % against +1.3 % for the in-place form with non-temporal loads and stores: not built.
// (STAN_OPT_VEC_STORE_NT: bit 0 = k_update stores p non-temporally, bit 1 = k_step stores r so.)
// The vector kernels' own loads -- operands a kernel only reads (v in k_step; r, x in k_update) and operands it rewrites
// in place (r in k_step, p in k_update) -- and the products' stores of y are non-temporal too.

namespace {

constexpr int VEC_BLOCKS = 2048;  // grid of the streaming vector kernels (8 blocks per CU; 1024 ... 16384 measured: profiles/r04/vector_grid_ab_in_cg.txt)
constexpr int VEC_T = 256;
constexpr int CHUNK = 32;         // iterations enqueued between two status polls
constexpr int CHUNK_DIST = 8;     // ... of a sharded loop

// device scalar slots (double)
enum { S_BNORM = 0, S_VMV = 1, S_R2NEW = 2, S_MERIT = 3, S_RHO0 = 4, S_RHO1 = 5, S_PMF0 = 6,
       S_PMF1 = 7, S_R2OUT = 8,
       // single-reduction loop: the three sums of one iteration are contiguous (ONE all-reduce)
       S_SR_GAMMA = 9, S_SR_DELTA = 10, S_SR_MERIT = 11, S_SR_GP0 = 12, S_SR_GP1 = 13, S_SR_AP0 = 14,
       S_SR_AP1 = 15,
       // fp64 check of a reduced-precision solve: ||b - A64 x||^2 and the merit sum of the same pass
       S_CHK_R2 = 16, S_CHK_MF = 17, S_NSCAL = 24 };
// device status slots (int64)
enum { T_ITER_A = 0, T_ITER_B = 1, T_TYPE = 2, T_ITERS = 3, T_XSEL = 4, T_NSTAT = 8 };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
// block sum (256 threads), valid in thread 0; fixed combination order
__device__ __forceinline__ double block_sum(double v, double *sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---- reductions folded into their producers ("last block done") --------------------------------
// Every block of a producing kernel leaves its partial sum(s) in `partial`, then takes a ticket;
// the block that draws the last ticket adds ALL partials in a fixed order (independent of which
// block that is: the result is bit-reproducible) and writes the scalar.  That removes the two
// one-block k_reduce launches per iteration from the stream (2 x (4.5 us + a kernel boundary) at
// 148^3; more where it matters: the sharded loop, whose per-rank kernels are 8 x shorter).
// Hand-off across XCDs (their L2s are not coherent, MI355X_MICROARCH.md "inter-workgroup
// visibility", first row of the table of measured forms): the partial is an agent-scope store
// (sc1, write-through), the storing lane waits for it (vmcnt(0)) before its agent-scope add to
// the one unsharded counter, the block whose add returned the last ticket reads every partial
// with agent-scope (sc1) loads after a workgroup barrier behind that add.
// STAN_OPT_CG_FOLD_REDUCE = 0 restores the separate k_reduce launches (same order: same bits).
__device__ __forceinline__ void st_agent(double *p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// r[j] = sum_i partial[i*NV + j] over np blocks by one 256-thread block, all NV sums in ONE pass
// over the partials (their loads overlap), fixed order; valid in thread 0
template <int NV>
__device__ __forceinline__ void sum_partials(const double *partial, int np, double *sh, double r[NV]) {
    constexpr int W = 16 / NV;   // loads in flight per thread: the last block's latency adds to the kernel
    double a[NV][W];
#pragma unroll
    for (int j = 0; j < NV; j++)
#pragma unroll
        for (int q = 0; q < W; q++) a[j][q] = 0;
    int i = threadIdx.x;
    for (; i + (W - 1) * 256 < np; i += W * 256) {
#pragma unroll
        for (int q = 0; q < W; q++)
#pragma unroll
            for (int j = 0; j < NV; j++) a[j][q] += ld_agent(partial + (int64_t)(i + q * 256) * NV + j);
    }
    for (; i < np; i += 256)
#pragma unroll
        for (int j = 0; j < NV; j++) a[j][0] += ld_agent(partial + (int64_t)i * NV + j);
#pragma unroll
    for (int j = 0; j < NV; j++) {
#pragma unroll
        for (int w = W / 2; w > 0; w >>= 1)   // fixed pairwise tree
#pragma unroll
            for (int q = 0; q < w; q++) a[j][q] += a[j][q + w];
        r[j] = block_sum(a[j][0], sh);
    }
}
// Tickets are two-level: block b first counts itself into sub-counter b % FOLD_SUB (a 128-B line
// of its own), the last arrival of a sub-counter counts that sub-counter into the top counter, the
// last arrival there finishes.  One flat counter cost k_step +10 us (rocprofv3, 148^3): its 2048
// blocks end together and 2048 adds to ONE address are served one after the other at the memory
// side; with 32 sub-counters the longest queue is 64.
constexpr int FOLD_SUB = 32;
constexpr int FOLD_LINE = 16;                               // uint64 per 128-B line
constexpr int FOLD_WORDS = (1 + FOLD_SUB) * FOLD_LINE;      // one counter set: top + sub-counters
struct fold_args {
    unsigned long long *counter;  // counter set (zero between kernels); nullptr: no fold
    unsigned nblocks;             // tickets this launch hands out (its grid size)
    int np;                       // partials to add (>= nblocks: earlier launches may have left some)
    double *out;                  // [NV] results
    p2p_out po;                   // sharded, peer to peer: the sums go to every rank's mailbox instead (p2p_device.h)
};
constexpr p2p_out NO_P2P = {nullptr, 0, 0, 0};
constexpr fold_args NO_FOLD = {nullptr, 0, 0, nullptr, NO_P2P};
// the finished sums r[0..NV) (valid in thread 0) to where the consumer will look for them
template <int NV>
__device__ __forceinline__ void publish_sums(double *out, const p2p_out &po, const double r[NV], double *sh) {
    if (po.pp) { p2p_publish<NV>(po, r, sh); return; }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < NV; j++) out[j] = r[j];  // read by the NEXT kernel: a plain store will do
    }
}
// Thread 0 of every block calls this after storing its partials with st_agent(); true (in every
// thread) for the block that arrived last.  `sh_last` is one int of LDS.
__device__ __forceinline__ bool fold_arrive(const fold_args &f, int *sh_last) {
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the partial has left this CU
        const unsigned sub = blockIdx.x % FOLD_SUB;
        const unsigned in_sub = (f.nblocks - sub + FOLD_SUB - 1) / FOLD_SUB;     // blocks b with b % SUB == sub
        const unsigned nsub = f.nblocks < (unsigned)FOLD_SUB ? f.nblocks : (unsigned)FOLD_SUB;
        int last = 0;
        unsigned long long t = __hip_atomic_fetch_add(f.counter + (1 + sub) * FOLD_LINE, 1ULL, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned long long)in_sub - 1) {
            t = __hip_atomic_fetch_add(f.counter, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = t == (unsigned long long)nsub - 1;
        }
        *sh_last = last;
    }
    __syncthreads();
    return *sh_last != 0;
}
template <int NV>
__device__ __forceinline__ void fold_finish(const fold_args &f, const double *partial, double *sh) {
    double r[NV];
    sum_partials<NV>(partial, f.np, sh, r);
    publish_sums<NV>(f.out, f.po, r, sh);
    // every ticket of this launch has been drawn: clear the set for the next one
    if (threadIdx.x <= FOLD_SUB)
        __hip_atomic_store(f.counter + threadIdx.x * FOLD_LINE, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A producing kernel that returns without doing its work (the solve has stopped; every rank takes the same
// decision) still owes its peers the arrival count of its reduction: their streams wait for it.
__device__ __forceinline__ void fold_skip(const fold_args &f) {
    if (f.counter && f.po.pp && f.po.signal && blockIdx.x == 0 && (int)threadIdx.x < f.po.pp->n)
        __hip_atomic_fetch_add(f.po.pp->sig_red[threadIdx.x][f.po.slot], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ bool stopped(const int64_t *st, int64_t k) {
    return st[T_ITER_A] < k || st[T_ITER_B] < k;
}

#include "cg_setup_kernels.inc"   // scaling (k_diag_scale, k_scale_matrix), value-stream copies (fp32, FIXED-48), the packed column stream (k_pack_cols), k_init / k_reduce / k_init_scalars

#include "spmv_kernels.inc"   // the products: k_spmv (one wavefront per slice), k_spmv_small (one workgroup per slice), k_spmv2 (two right-hand sides), k_spmv_fold (folded rows)

#include "cg_vector_kernels.inc"   // the vector kernels of the CG loop: k_step, k_refresh, k_update, k_vec_sr (single-reduction form), result / expand / compress

inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }
inline unsigned vec_grid(int64_t n) {
    int64_t b = (n + VEC_T - 1) / VEC_T;
    if (b < 1) b = 1;
    return (unsigned)(b > VEC_BLOCKS ? VEC_BLOCKS : b);
}

struct dev_bufs {
    stan_ctx *ctx = nullptr;
    std::vector<void *> p;
    ~dev_bufs() {
        for (void *q : p) stan_dfree(ctx, q);
    }
};
template <typename T>
int alloc(stan_ctx *ctx, dev_bufs &b, T **p, size_t n) {
    b.ctx = ctx;
    int rc = stan_dmalloc(ctx, p, n);
    if (rc == STAN_OK) b.p.push_back((void *)*p);
    return rc;
}

// k_spmv_small instead of k_spmv: decided by the GLOBAL number of block rows, so that a shard and the
// whole matrix sum their rows in the same order
inline bool stan_small_system(const stan_ctx *ctx, const stan_matrix *K) {
    return ctx->spmv_variant < 0 && K->nb_glob <= ctx->spmv_small_rows;
}

inline bool stan_pair_kernel(const stan_ctx *ctx) { return ctx->spmv_variant == 20; }

// the folded form of the value stream `vals` of K, if the products are to read it (fold.hip)
template <typename VT> const VT *fold_vals(const stan_ctx *ctx, const stan_matrix *K, const VT *vals);
template <> const double *fold_vals<double>(const stan_ctx *ctx, const stan_matrix *K, const double *vals) {
    if (ctx->fold_probe && (const void *)vals == ctx->fold_probe) return vals;
    return ctx->row_folding != 0 && vals == K->d_vals ? K->d_fold_vals : nullptr;
}
template <> const float *fold_vals<float>(const stan_ctx *ctx, const stan_matrix *K, const float *vals) {
    if (ctx->fold_probe && (const void *)vals == ctx->fold_probe) return vals;
    return ctx->row_folding != 0 && vals == K->d_vals32 ? K->d_fold_vals32 : nullptr;
}
template <> const uint32_t *fold_vals<uint32_t>(const stan_ctx *ctx, const stan_matrix *K, const uint32_t *vals) {
    if (ctx->fold_probe && (const void *)vals == ctx->fold_probe) return vals;
    return ctx->row_folding != 0 && vals == K->d_vals48 ? K->d_fold_vals48 : nullptr;
}
inline colstream fold_cols_of(const stan_ctx *ctx, const stan_matrix *K) {
    return ctx->cols16 && K->d_fold_cols16 ? make_colstream(K->d_fold_cols16, K->d_fold_colbase, K->d_fold_pair_ptr, K->d_fold_packed, K->nfslots)
                                           : NO_COLSTREAM;
}
// which: 0 = all slices, 1 = interior list, 2 = boundary list (partials offset by the
// interior launch's block count).  Returns the number of partial slots this launch writes.
// `fold`: counter/out of the folded reduction (counter == nullptr: partials only); nblocks and np
// are filled in here (np = this launch's slots + poff: the boundary launch of a split product
// also adds up what the interior launch left).
template <typename VT, int DOT>
unsigned launch_spmv(stan_ctx *ctx, stan_matrix *K, const VT *vals, const double *x, double *y,
                     double *partial, const int64_t *st, int64_t k, int which = 0,
                     hipStream_t stream = nullptr, fold_args fold = NO_FOLD) {
    if (!stream) stream = ctx->stream;
    const int32_t *slist = which == 1 ? K->d_sl_int : which == 2 ? K->d_sl_bnd : nullptr;
    const int32_t nlist = which == 1 ? K->n_sl_int : which == 2 ? K->n_sl_bnd : K->nslices;
    const colstream cs = ctx->cols16 && K->d_cols16 ? make_colstream(K->d_cols16, K->d_colbase, K->d_pair_ptr, K->d_slice_packed, K->nslots)
                                                    : NO_COLSTREAM;
    if (stan_small_system(ctx, K)) {   // one workgroup per slice (k_spmv_small); partials per slice
        const int32_t poff_s = which == 2 ? K->n_sl_int : 0;
        const unsigned grid_s = (unsigned)nlist;
        if (grid_s == 0) return 0;
        fold.nblocks = grid_s;
        fold.np = (int)grid_s + poff_s;
        hipLaunchKernelGGL((k_spmv_small<VT, DOT>), dim3(grid_s), dim3(256), 0, stream, K->nslices, K->nloc,
                           K->d_slot_ptr, K->d_rowof, K->d_cols, vals, x, y, partial, st, k, slist, nlist, poff_s, fold, cs);
        return grid_s;
    }
    if (stan_pair_kernel(ctx)) {   // two wavefronts per slice, two slices per workgroup (k_spmv_pair; STAN_OPT_SPMV_VARIANT 20)
        const int32_t poff_p = which == 2 ? (int32_t)nblk(K->n_sl_int, 2) : 0;
        const unsigned grid_p = nblk(nlist, 2);
        if (grid_p == 0) return 0;
        fold.nblocks = grid_p;
        fold.np = (int)grid_p + poff_p;
        hipLaunchKernelGGL((k_spmv_pair<VT, DOT, true>), dim3(grid_p), dim3(256), 0, stream, K->nslices, K->nloc,
                           K->d_slot_ptr, K->d_rowof, K->d_cols, vals, x, y, partial, st, k, slist, nlist, poff_p, fold, cs);
        return grid_p;
    }
    const int32_t poff = which == 2 ? (int32_t)nblk(K->n_sl_int, 4) : 0;
    const unsigned grid = nblk(nlist, 4);
    if (grid == 0) return 0;
    fold.nblocks = grid;
    fold.np = (int)grid + poff;
    if (const VT *fv = fold_vals<VT>(ctx, K, vals)) {
        hipLaunchKernelGGL((k_spmv_fold<VT, DOT, 1>), dim3(grid), dim3(256), 0, stream, K->nslices, K->nloc, K->d_fold_ptr,
                           K->d_rowof, K->d_fold_meta, K->d_fold_cols, fv, x, (const double *)nullptr, y, (double *)nullptr,
                           partial, st, k, slist, nlist, poff, fold, fold_cols_of(ctx, K));
        return grid;
    }
#define SPMV_CASE(V)                                                                          \
    case V:                                                                                   \
        hipLaunchKernelGGL((k_spmv<VT, DOT, V>), dim3(grid), dim3(256), 0, stream, K->nslices, \
                           K->nloc, K->d_slot_ptr, K->d_rowof, K->d_cols, vals, x, y, partial, st, k, \
                           slist, nlist, poff, fold, cs);                                     \
        break;
    // auto (-1): non-temporal matrix stream + XCD-chunked workgroup mapping (variant 9), with the
    // loop unrolled by 4 for the FIXED-48 stream, whose iterations carry 21 % fewer bytes in
    // flight (variant 12).  One process, one box (tools/fx48_variants.py, min of 3 x 20 launches):
    // fp64 plain 1.181 / nt 1.102 / nt+chunk 1.100 ms; FIXED-48 1.003 / 0.955 / nt+unroll4 0.925;
    // fp32 0.636 / 0.583 / 0.580.
    const int variant = ctx->spmv_variant >= 0 ? ctx->spmv_variant : vstream<VT>::FX ? 12 : 9;
    switch (variant) {
        SPMV_CASE(9) SPMV_CASE(12)
        default:
        SPMV_CASE(0)
    }
#undef SPMV_CASE
    return grid;
}

template <typename VT>
unsigned launch_spmv2(stan_ctx *ctx, stan_matrix *K, const VT *vals, const double *x, const double *x2,
                      double *y, double *y2, double *partial, const int64_t *st, int64_t k, int which,
                      hipStream_t stream, fold_args fold) {
    const int32_t *slist = which == 1 ? K->d_sl_int : which == 2 ? K->d_sl_bnd : nullptr;
    const int32_t nlist = which == 1 ? K->n_sl_int : which == 2 ? K->n_sl_bnd : K->nslices;
    const int32_t poff = which == 2 ? (int32_t)nblk(K->n_sl_int, 4) : 0;
    const unsigned grid = nblk(nlist, 4);
    if (grid == 0) return 0;
    fold.nblocks = grid;
    fold.np = (int)grid + poff;
    if (const VT *fv = fold_vals<VT>(ctx, K, vals)) {
        hipLaunchKernelGGL((k_spmv_fold<VT, 1, 2>), dim3(grid), dim3(256), 0, stream, K->nslices, K->nloc, K->d_fold_ptr,
                           K->d_rowof, K->d_fold_meta, K->d_fold_cols, fv, x, x2, y, y2, partial, st, k, slist, nlist, poff, fold,
                           fold_cols_of(ctx, K));
        return grid;
    }
    const colstream cs = ctx->cols16 && K->d_cols16 ? make_colstream(K->d_cols16, K->d_colbase, K->d_pair_ptr, K->d_slice_packed, K->nslots)
                                                    : NO_COLSTREAM;
    hipLaunchKernelGGL((k_spmv2<VT>), dim3(grid), dim3(256), 0, stream, K->nslices, K->nloc, K->d_slot_ptr,
                       K->d_rowof, K->d_cols, vals, x, x2, y, y2, partial, st, k, slist, nlist, poff, fold, cs);
    return grid;
}

// value-stream dispatch of one product (DOT as in k_spmv)
template <int DOT>
unsigned launch_spmv_any(stan_ctx *ctx, stan_matrix *K, int stream_kind, const double *x, double *y,
                         double *partial, const int64_t *st, int64_t k, int which, hipStream_t s,
                         fold_args fold) {
    if (stream_kind == STAN_PREC_MIXED)
        return launch_spmv<float, DOT>(ctx, K, K->d_vals32, x, y, partial, st, k, which, s, fold);
    if (stream_kind == STAN_PREC_FIXED48)
        return launch_spmv<uint32_t, DOT>(ctx, K, K->d_vals48, x, y, partial, st, k, which, s, fold);
    return launch_spmv<double, DOT>(ctx, K, K->d_vals, x, y, partial, st, k, which, s, fold);
}

}  // namespace

// Vectors of the CG for a matrix of K's sizes, owned by the context (see stan_cg_ws).
int stan_cg_workspace(stan_ctx *ctx, const stan_matrix *K) {
    const int64_t npad = (int64_t)K->nslices * 64;
    const int64_t ng = 3 * ((npad > K->nloc + K->nhalo ? npad : K->nloc + K->nhalo));
    const int64_t n3 = 3 * K->nloc > 0 ? 3 * K->nloc : 1;
    stan_cg_ws &ws = ctx->ws;
    if (ws.p && ws.ng >= ng && ws.n3 >= n3 && ws.ng <= ng + ng / 2 + 64) return STAN_OK;
    stan_cg_workspace_free(ctx);
    for (double **q : {&ws.xb[0], &ws.xb[1], &ws.p, &ws.r}) STANCHK(stan_dmalloc(ctx, q, (size_t)(ng > 0 ? ng : 1)));
    for (double **q : {&ws.v, &ws.w, &ws.bh, &ws.sv}) STANCHK(stan_dmalloc(ctx, q, (size_t)n3));
    ws.ng = ng; ws.n3 = n3;
    return STAN_OK;
}
void stan_cg_workspace_free(stan_ctx *ctx) {
    stan_cg_ws &ws = ctx->ws;
    if (ws.vw_owner) {   // v, w carved out of one block (placement.hip, second stage): plain device memory, not the pool's
        stan_dfree(ctx, ws.vw_owner);
        ws.vw_owner = ws.v = ws.w = nullptr;
        ctx->prof_placement_moved_vectors = 0;   // the next matrix of another size searches afresh
    }
    for (double **q : {&ws.xb[0], &ws.xb[1], &ws.p, &ws.r, &ws.v, &ws.w, &ws.bh, &ws.sv}) {
        if (*q) stan_dfree(ctx, *q);
        *q = nullptr;
    }
    ws.ng = ws.n3 = 0;
}

// Last resort of the placement search (placement.hip): no value candidate was clear of the vectors'
// group, and all of them are still allocated -- so NEW vectors, allocated now, lie beyond them.
// They are taken straight from the driver (a parked pool block would be the old memory again); the
// caller probes with them and keeps them (commit) or not.
int stan_cg_workspace_move(stan_ctx *ctx, const stan_matrix *K, bool commit, stan_cg_ws *saved) {
    stan_cg_ws &ws = ctx->ws;
    if (saved && !commit && saved->p == nullptr) {          // step 1: swap fresh vectors in, keep the old ones in *saved
        if (!ws.p) return STAN_OK;
        *saved = ws;
        stan_cg_ws nw;   // (nw.vw_owner stays null: eight blocks of their own)
        nw.ng = ws.ng; nw.n3 = ws.n3;
        bool ok = true;
        double **dst[8] = {&nw.xb[0], &nw.xb[1], &nw.p, &nw.r, &nw.v, &nw.w, &nw.bh, &nw.sv};
        for (int i = 0; i < 8 && ok; i++)
            ok = hipMalloc((void **)dst[i], (size_t)(i < 4 ? (nw.ng > 0 ? nw.ng : 1) : nw.n3) * 8) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            for (double **q : dst) if (*q) hipFree(*q);
            saved->p = nullptr;
            return STAN_OK;   // no memory for it: keep what we have
        }
        ws = nw;
        (void)K;
        return STAN_OK;
    }
    if (!saved || saved->p == nullptr) return STAN_OK;
    // step 2: keep the new vectors (release the old ones for real) or go back to the old ones
    stan_cg_ws &drop = commit ? *saved : ws;
    stan_cg_ws keep = commit ? ws : *saved;
    if (drop.vw_owner) { ctx->pool.live.erase((void *)drop.vw_owner); hipFree(drop.vw_owner); drop.vw_owner = drop.v = drop.w = nullptr; }
    for (double *q : {drop.xb[0], drop.xb[1], drop.p, drop.r, drop.v, drop.w, drop.bh, drop.sv}) {
        if (!q) continue;
        ctx->pool.live.erase((void *)q);
        hipFree(q);
    }
    ws = keep;
    if (commit && ctx->pool.enabled)   // the new blocks are pooled-class blocks from now on
        for (double *q : {ws.xb[0], ws.xb[1], ws.p, ws.r}) if ((size_t)ws.ng * 8 >= stan_pool::MIN_BYTES) ctx->pool.live[(void *)q] = (size_t)ws.ng * 8;
    if (commit && ctx->pool.enabled)
        for (double *q : {ws.vw_owner ? (double *)nullptr : ws.v, ws.vw_owner ? (double *)nullptr : ws.w, ws.bh, ws.sv})
            if (q && (size_t)ws.n3 * 8 >= stan_pool::MIN_BYTES) ctx->pool.live[(void *)q] = (size_t)ws.n3 * 8;
    saved->p = nullptr;
    return STAN_OK;
}

// Round 4 (tools/lab/spmv_steps_lab.cpp, profiles/r04/spmv_steps/): what makes a pairing slow is the vector the product
// WRITES lying in the memory group of the values it reads -- the gather vector's place does not matter.  So the cheap move is
// the two product buffers v and w alone, CARVED out of a block the caller has placed (a fresh small allocation would land in
// whatever hole the allocator finds, not behind the caller's spacers).  set: v, w point into `block` (saved[] takes the old
// v, w, owner) or, with block == nullptr, back to saved[]; adopt: the carved pair stays, `block` becomes its owner and the
// saved pair is released.
static int64_t vw_stride(const stan_cg_ws &ws) { return ((ws.n3 > 0 ? ws.n3 : 1) + 511) & ~(int64_t)511; }
size_t stan_cg_products_bytes(const stan_ctx *ctx) { return (size_t)(2 * vw_stride(ctx->ws)) * 8; }
void stan_cg_products_set(stan_ctx *ctx, double *block, double *saved[3]) {
    stan_cg_ws &ws = ctx->ws;
    if (block) {
        saved[0] = ws.v; saved[1] = ws.w; saved[2] = ws.vw_owner;
        ws.v = block; ws.w = block + vw_stride(ws);
    } else {
        ws.v = saved[0]; ws.w = saved[1]; ws.vw_owner = saved[2];
    }
}
void stan_cg_products_adopt(stan_ctx *ctx, double *block, size_t bytes, double *saved[3]) {
    stan_cg_ws &ws = ctx->ws;
    if (saved[2]) { ctx->pool.live.erase((void *)saved[2]); hipFree(saved[2]); }
    else for (int i = 0; i < 2; i++) if (saved[i]) { ctx->pool.live.erase((void *)saved[i]); hipFree(saved[i]); }
    ws.v = block; ws.w = block + vw_stride(ws); ws.vw_owner = block;
    // NOT a pooled block: 1-4 GB that hold 2 x n3 doubles would sit in the pool for the context's life once
    // the workspace is re-sized, matching no later request; stan_cg_workspace_free gives it straight back to the driver
    (void)bytes;
    saved[0] = saved[1] = saved[2] = nullptr;
}

// Diagonal scaling of the matrix (once per matrix): A^ = S K S.  First the vector s (ensure_scale_vector), then the values:
// by a pass of its own (ensure_scaled), or -- the fp64 loop on one rank -- by the loop's first product (k_spmv_first,
// cg_run::iterate), which marks the matrix scaled itself.
static int ensure_scale_vector(stan_ctx *ctx, stan_matrix *K) {
    const int64_t npad = (int64_t)K->nslices * 64;
    const int64_t ns = 3 * (npad + K->nhalo);
    if (!K->d_scale) STANCHK(stan_dmalloc(ctx, &K->d_scale, (size_t)ns));
    hipLaunchKernelGGL(k_fill, dim3(vec_grid(ns)), dim3(VEC_T), 0, ctx->stream, K->d_scale, ns, 1.0);
    if (K->nloc > 0)
        hipLaunchKernelGGL(k_diag_scale, dim3(nblk(K->nloc, 256)), dim3(256), 0, ctx->stream, K->nloc,
                           K->d_rowlen, K->d_posof, K->d_slot_ptr, K->d_cols, K->d_vals, K->d_scale);
    if (ctx->comm || ctx->nranks > 1) {
        if (ctx->comm_p2p && ctx->p2p) {
            // Peer to peer, my neighbours write their rows straight into my halo region, which the k_fill above
            // has just initialised: nobody may write before everybody has done that.  One empty reduction
            // (every rank counts itself into every rank's counter, every stream waits for all) orders it; the
            // exchanges inside the CG are ordered by the loop's own reductions.  (Found with the ranks in
            // separate processes, where mapping the peers' vectors delays some ranks by milliseconds: a
            // neighbour's scaling factors arrived before the fill and were overwritten with 1.0.)
            const p2p_out po{stan_p2p_table(ctx), stan_p2p_reduce_slot(ctx), 3, 1};
            hipLaunchKernelGGL(k_reduce<1>, dim3(1), dim3(256), 0, ctx->stream, (const double *)nullptr, 0, (double *)nullptr, po, (const int64_t *)nullptr, (int64_t)0);
            STANCHK(stan_p2p_reduce_wait(ctx));
        }
        STANCHK(stan_comm_halo_exchange(ctx, K, K->d_scale));
    }
    HIPCHK(ctx, hipGetLastError());
    return STAN_OK;
}
// what changes for the rest of the library once the values carry S K S
static void mark_scaled(stan_ctx *ctx, stan_matrix *K) {
    K->scaled = true;
    // copies of the value stream made before the scaling (stan_hip_spmv_bench on a fresh matrix)
    // hold the unscaled K: a later solve must not iterate on them
    if (K->d_vals32) { stan_dfree(ctx, K->d_vals32); K->d_vals32 = nullptr; }
    if (K->d_vals48) { stan_dfree(ctx, K->d_vals48); K->d_vals48 = nullptr; }
    stan_matrix_drop_folded_values(ctx, K);
    K->fx48_refused = false;
}
static int ensure_scaled(stan_ctx *ctx, stan_matrix *K) {
    if (K->scaled) return STAN_OK;
    STANCHK(ensure_scale_vector(ctx, K));
    if (K->nslices > 0)
        hipLaunchKernelGGL(k_scale_matrix, dim3(nblk(K->nslices, 4)), dim3(256), 0, ctx->stream,
                           K->nslices, K->d_slot_ptr, K->d_rowof, K->d_cols, K->d_vals, K->d_scale);
    HIPCHK(ctx, hipGetLastError());
    mark_scaled(ctx, K);
    return STAN_OK;
}

int stan_matrix_make_fp32(stan_ctx *ctx, stan_matrix *K) {
    if (K->d_vals32) return STAN_OK;
    const int64_t n = K->nslots * 9 * 64;
    STANCHK(stan_dmalloc_streamed(ctx, (void **)&K->d_vals32, (size_t)n * 4,
                                  [&](const void *q, float *ms, bool self) {
                                      return stan_spmv_probe(ctx, K, q, (size_t)n * 4, STAN_PREC_MIXED, ms, self);
                                  }));
    hipLaunchKernelGGL(k_to_fp32, dim3(vec_grid(n) * 4), dim3(VEC_T), 0, ctx->stream, K->d_vals,
                       K->d_vals32, n);
    HIPCHK(ctx, hipGetLastError());
    return STAN_OK;
}

// FIXED-48 copy of the scaled values.  Returns STAN_OK with K->d_vals48 == nullptr when some
// entry is not representable (K not SPD): the caller then streams the fp64 values.
int stan_matrix_make_fx48(stan_ctx *ctx, stan_matrix *K) {
    if (K->d_vals48 || K->fx48_refused) return STAN_OK;
    if (K->nslots == 0) return STAN_OK;
    uint32_t *out;
    STANCHK(stan_dmalloc_streamed(ctx, (void **)&out, (size_t)K->nslots * 14 * 64 * 4,
                                  [&](const void *q, float *ms, bool self) {
                                      return stan_spmv_probe(ctx, K, q, (size_t)K->nslots * 14 * 64 * 4, STAN_PREC_FIXED48, ms, self);
                                  }));
    unsigned long long *d_bad = (unsigned long long *)(ctx->d_status + SS_COUNTER);
    HIPCHK(ctx, hipMemsetAsync(d_bad, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_to_fx48, dim3((unsigned)nblk(K->nslots * 64, 256)), dim3(256), 0, ctx->stream,
                       K->nslots, K->d_vals, out, d_bad);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_status + SS_COUNTER, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->h_status[SS_COUNTER] != 0) { stan_dfree(ctx, out); K->fx48_refused = true; return STAN_OK; }
    K->d_vals48 = out;
    return STAN_OK;
}

// Packed column stream of K (struct colstream), built once per matrix; the int32 columns stay (the
// assembly, scaling and export kernels use them).
int stan_matrix_make_cols16(stan_ctx *ctx, stan_matrix *K) {
    if (K->d_cols16 || K->nslices <= 0) return STAN_OK;
    return stan_pack_columns(ctx, K->nslices, K->nslots, K->d_slot_ptr, K->d_cols, &K->d_cols16, &K->d_colbase, &K->d_pair_ptr,
                             &K->d_slice_packed, &K->slots_packed, K->nloc, K->d_rowof, K->d_rowlen, &K->slots_packed2);
}
// the same for any sliced column stream (the folded copy of fold.hip has its own); *packed stays nullptr when the
// pair index would not fit an int32
int stan_pack_columns(stan_ctx *ctx, int32_t nslices, int64_t nslots, const int32_t *d_slot_ptr, const int32_t *d_cols,
                      uint32_t **packed_out, int32_t **base_out, int32_t **pair_ptr_out, uint8_t **ok_out, int64_t *slots_packed,
                      int64_t nloc, const int32_t *d_rowof, const int32_t *d_rowlen, int64_t *slots_packed2) {
    hipStream_t st_ = ctx->stream;
    dev_bufs bufs;
    int32_t *cnt; int64_t *ptr64;
    STANCHK(alloc(ctx, bufs, &cnt, (size_t)nslices + 1));
    STANCHK(alloc(ctx, bufs, &ptr64, (size_t)nslices + 2));
    hipLaunchKernelGGL(k_pair_counts, dim3(nblk(nslices, 256)), dim3(256), 0, st_, nslices, d_slot_ptr, cnt);
    STANCHK(stan_scan_exclusive(ctx, cnt, ptr64, nslices));
    std::vector<int64_t> h((size_t)nslices + 1);
    HIPCHK(ctx, hipMemcpyAsync(h.data(), ptr64, h.size() * 8, hipMemcpyDeviceToHost, st_));
    HIPCHK(ctx, hipStreamSynchronize(st_));
    const int64_t npairs = h[(size_t)nslices];
    if (npairs >= ((int64_t)1 << 31)) return STAN_OK;   // pair index is int32: keep the plain columns
    std::vector<int32_t> h32(h.size());
    for (size_t i = 0; i < h.size(); i++) h32[i] = (int32_t)h[i];
    STANCHK(stan_dmalloc(ctx, pair_ptr_out, h32.size()));
    // one allocation: [n] base, [n] base2, [nslices] cmask (64-bit words), n = max(nslots, 1): make_colstream
    const size_t nb_ = (size_t)(nslots > 0 ? nslots : 1);
    STANCHK(stan_dmalloc(ctx, base_out, 2 * nb_ + 2 * (size_t)nslices + 2));
    STANCHK(stan_dmalloc(ctx, ok_out, (size_t)nslices));
    uint32_t *packed;
    STANCHK(stan_dmalloc(ctx, &packed, (size_t)(npairs > 0 ? npairs : 1) * 64));
    HIPCHK(ctx, hipMemcpyAsync(*pair_ptr_out, h32.data(), h32.size() * 4, hipMemcpyHostToDevice, st_));
    int32_t *b2_ = *base_out + nb_;
    unsigned long long *cm_ = (unsigned long long *)(*base_out + 2 * nb_);
    if (((uintptr_t)cm_ & 7) != 0) cm_ = (unsigned long long *)((uintptr_t)cm_ + 4);   // (never: 2 n ints from an aligned block)
    hipLaunchKernelGGL(k_pack_cols, dim3(nblk(nslices, 4)), dim3(256), 0, st_, nslices, nloc, d_slot_ptr, d_cols, d_rowof, d_rowlen,
                       *pair_ptr_out, packed, *base_out, b2_, cm_, *ok_out);
    unsigned long long *d_cnt = (unsigned long long *)(ctx->d_status + SS_COUNTER);   // two words: SS_COUNTER, SS_H_ERRCOPY
    HIPCHK(ctx, hipMemsetAsync(d_cnt, 0, 16, st_));
    hipLaunchKernelGGL(k_count_ok, dim3(nblk(nslices, 256)), dim3(256), 0, st_, nslices, *ok_out, d_slot_ptr, d_cnt);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_status + SS_COUNTER, d_cnt, 16, hipMemcpyDeviceToHost, st_));
    HIPCHK(ctx, hipStreamSynchronize(st_));   // h32 must outlive the copy
    *slots_packed = ctx->h_status[SS_COUNTER];
    if (slots_packed2) *slots_packed2 = ctx->h_status[SS_COUNTER + 1];
    *packed_out = packed;
    return STAN_OK;
}

// NOTE on the halo layout: vectors that are gathered by the SpMV (p, x) hold the owned
// block rows first, padded to whole slices, then the halo block columns:
//   [ 3*nslices*64 owned+pad | 3*nhalo ]
// Local column indices >= nloc written by the symbolic phase are relative to nloc, so
// the halo region must start at 3*nloc: the pad only exists for nranks == 1 tails, where
// nhalo == 0.  For nranks > 1 every rank's row count is a multiple of 64 except the last
// rank's; the gather vectors are therefore sized 3*(max(nloc, pad) + nhalo) and the halo
// always sits at 3*nloc.
// Host waits of the loop.  Without peer-to-peer exchanges: the plain blocking calls.  With them the stream may sit in a
// wait for a peer that is gone, and a blocking call would never return (the polling wavefront keeps the queue busy):
// the host polls instead, and after stan_p2p_stall_seconds() without completion releases this rank's waits
// (stan_p2p_release_own), lets the queue drain and reports STAN_E_COMM.  `ev` == nullptr: the whole stream.
static int cg_wait(stan_ctx *ctx, bool p2p, hipStream_t st, hipEvent_t ev) {
    if (!p2p) {
        if (ev) HIPCHK(ctx, hipEventSynchronize(ev));
        else HIPCHK(ctx, hipStreamSynchronize(st));
        return STAN_OK;
    }
    const double bound = stan_p2p_stall_seconds();
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    for (;;) {
        const hipError_t e = ev ? hipEventQuery(ev) : hipStreamQuery(st);
        if (e == hipSuccess) return STAN_OK;
        if (e != hipErrorNotReady) { ctx->err = std::string("cg: ") + hipGetErrorString(e); (void)hipGetLastError(); return STAN_E_HIP; }
        (void)hipGetLastError();
        if (++spins > 200) std::this_thread::sleep_for(std::chrono::microseconds(spins > 2000 ? 500 : 50));
        if (ctx->p2p->broken.load()) break;   // somebody else (run_all, another rank's release) gave up already
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > bound) break;
    }
    stan_p2p_release_own(ctx);
    (void)hipStreamSynchronize(st);   // the released waits pass: what was enqueued runs out
    (void)hipGetLastError();
    ctx->err = "cg: peer-to-peer exchange made no progress (a peer rank failed or never arrived); this rank's waits were released";
    return STAN_E_COMM;
}

namespace {

// ---- one solve -------------------------------------------------------------------------------------------
// cg_run holds the state of ONE call of stan_cg_device and is its three parts:
//   setup()    streams and vectors: workspace, scaling, packed columns, value streams, scalars, counters
//   pass()     ONE run of the CG loop on the right-hand side in bh (begin_pass: k_init / k_init_b, first
//              residual test; iterate: chunks of iterations enqueued ahead of a polled status word)
//   finish()   U = S x^, the report, the profile
// and, for the reduced-precision value streams (STAN_PREC_MIXED, STAN_PREC_FIXED48), what lies between two passes:
//   fp64_check()   r_t = b^ - A^64 x^ with the fp64 values (which stay resident next to their copy): the residual
//                  the caller gets REPORTED, and the right-hand side of the next pass when it misses eps
//                  (STAN_OPT_CG_REFINE: iterative refinement; the passes' iterates add up in fp64).
// The fp64 stream makes exactly one pass and no check: alglib's loop as the reference runs it.
struct cg_run {
    stan_ctx *ctx;
    stan_matrix *K;
    const double *d_F;
    double eps_f;
    int32_t max_its;
    int32_t precision_mode;
    double *d_U;

    hipStream_t st_ = nullptr;
    event_bag events;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool dist = false, p2p = false, sr = false, foldr = false, split = false;
    int vs = STAN_PREC_FP64;          // the stream the loop's products read on THIS rank (FIXED-48 falls back to fp64 on a rank that owns no
                                      // rows or whose shard is not representable: never a base for decisions the ranks must share)
    bool lazy_scale = false;          // the values are still K: the loop's first product scales them (k_spmv_first)
    bool reduced = false;             // the caller asked for a reduced-precision stream: fp64 check + refinement (every rank alike)
    int refine = 0;                   // STAN_OPT_CG_REFINE, reduced-precision modes only
    int64_t n3 = 0, npad = 0, ng = 0, dof0 = 0;
    dev_bufs bufs;
    double *xb[2] = {nullptr, nullptr}, *p = nullptr, *r = nullptr, *v = nullptr, *w = nullptr, *bh = nullptr;
    double *partial = nullptr, *sc = nullptr, *sv = nullptr;
    double *xacc = nullptr, *b0 = nullptr;   // refinement: the sum of the passes' iterates, the original b^ (allocated when a second pass starts)
    int64_t *stt = nullptr;
    unsigned long long *tick = nullptr;
    const stan_p2p_dev *p2p_tab = nullptr;
    unsigned vg = 1;
    int64_t its_before_restart = 1;
    static constexpr int64_t hard_cap = 0x7fffffff;   // iteration counter is int32 in the report
    int64_t *h_st = nullptr;          // pinned status words
    double *h_sc = nullptr;           // pinned copy of the scalars
    hipEvent_t poll[2] = {nullptr, nullptr};
    // profile
    std::vector<hipEvent_t> red_ev, halo_ev, spmv_ev, spmv2_ev, spmv64_ev;
    std::vector<int64_t> spmv_k, spmv2_k;   // iteration of each timed launch
    int64_t n_coll = 0, n_wait = 0, n_launch = 0, n_enqueued = 0;
    double prof_spmv_ms = 0, prof_spmv2_ms = 0, prof_spmv64_ms = 0;
    int64_t prof_spmv_n = 0, prof_spmv2_n = 0, prof_spmv64_n = 0;
    // result of the last pass
    int pass_type = 0;
    int64_t pass_its = 0;
    const double *xfin = nullptr;

    // Where the sums of a reduction go.  One rank, or RCCL: the local scalars (RCCL all-reduces them in
    // place).  Peer to peer: every rank's mailbox slot `slot`, columns j0.. (+ arrival count when this is
    // the producer that completes the exchange); the consumers then read through a red_src.
    p2p_out p2p_to(int j0, bool signal) {
        return p2p ? p2p_out{p2p_tab, stan_p2p_reduce_slot(ctx), j0, signal ? 1 : 0} : NO_P2P;
    }
    // folded reductions: counter set 0 serves the products, set 1 the vector kernels
    fold_args fold_to(int which, double *out, p2p_out po = NO_P2P) {
        return fold_args{foldr ? tick + FOLD_WORDS * which : nullptr, 0, 0, out, po};
    }
    fold_args vec_fold(double *out, p2p_out po = NO_P2P) {
        fold_args f = fold_to(1, out, po);
        f.nblocks = vg; f.np = (int)vg;
        return f;
    }
    void reduce_if_unfolded(int np, int nv, double *out, p2p_out po = NO_P2P, int64_t k_ = -1) {
        if (foldr || np <= 0) return;
        const int64_t *sk = k_ >= 1 ? stt : nullptr;   // (k_init's sum is formed before the status exists)
        if (nv == 2) hipLaunchKernelGGL(k_reduce<2>, dim3(1), dim3(256), 0, st_, partial, np, out, po, sk, k_);
        else hipLaunchKernelGGL(k_reduce<1>, dim3(1), dim3(256), 0, st_, partial, np, out, po, sk, k_);
    }
    // One exchange point of the sharded loop: RCCL all-reduce of `count` scalars in place, or the stream
    // wait for every rank's arrival; returns where the consumers find the sums.
    int exchange_sums(double *scalars, int count, red_src *rs) {
        *rs = red_src{nullptr, 0, nullptr, 0};
        if (!dist) return STAN_OK;
        if (ctx->profiling) { red_ev.push_back(events.make()); hipEventRecord(red_ev.back(), st_); }
        int rc_ = STAN_OK;
        if (p2p) {
            *rs = red_src{stan_p2p_mailbox(ctx, stan_p2p_reduce_slot(ctx)), ctx->nranks, nullptr, 0};
            rc_ = stan_p2p_reduce_wait(ctx, &rs->ctr, &rs->want);   // (wait mode 2: the consumers poll rs->ctr themselves)
            if (!rs->ctr) n_wait++;
        } else {
            rc_ = stan_comm_allreduce_sum_f64(ctx, scalars, (size_t)count);
            n_coll++;
        }
        if (ctx->profiling) { red_ev.push_back(events.make()); hipEventRecord(red_ev.back(), st_); }
        return rc_;
    }
    int halo(double *x) {
        if (ctx->profiling) { halo_ev.push_back(events.make()); hipEventRecord(halo_ev.back(), st_); }
        const int rc_ = stan_comm_halo_exchange(ctx, K, x);
        if (p2p && !K->nbr.empty()) n_wait++;
        if (ctx->profiling) { halo_ev.push_back(events.make()); hipEventRecord(halo_ev.back(), st_); }
        return rc_;
    }

    // y = A^ x (x gets its halo filled first when sharded) with `dot` sums (k_spmv's DOT) reduced into out[0..dot):
    // folded into the last launch of the product, or by k_reduce.  kind: the value stream (the loop's own, or
    // STAN_PREC_FP64 for the check / refresh products of a reduced-precision solve).
    int spmv(double *x, double *y, int dot, double *out, int64_t k, p2p_out po, int kind, bool extra = false) {
        const bool own = !extra;   // extra: an fp64 product inside a reduced-precision solve (check, refresh): timed apart
        if (ctx->profiling) {
            hipEvent_t a = events.make(), b = events.make();
            hipEventRecord(a, st_);
            if (own) { spmv_ev.push_back(a); spmv_ev.push_back(b); spmv_k.push_back(k); }
            else { spmv64_ev.push_back(a); spmv64_ev.push_back(b); }
        }
        auto go = [&](int which, hipStream_t s, bool last) -> unsigned {
            fold_args f = (dot && last) ? fold_to(0, out, po) : NO_FOLD;
            n_launch++;
            if (lazy_scale) {   // the first product of this matrix (one rank, fp64 stream, all slices): scale on the way
                lazy_scale = false;
                const unsigned grid = nblk(K->nslices, 4);
                f.nblocks = grid; f.np = (int)grid;
                const colstream cs = ctx->cols16 && K->d_cols16 ? make_colstream(K->d_cols16, K->d_colbase, K->d_pair_ptr, K->d_slice_packed, K->nslots) : NO_COLSTREAM;
                if (dot == 2) hipLaunchKernelGGL(k_spmv_first<2>, dim3(grid), dim3(256), 0, s, K->nslices, K->nloc, K->d_slot_ptr, K->d_rowof, K->d_cols, K->d_vals, K->d_scale, x, y, partial, stt, k, f, cs);
                else if (dot == 1) hipLaunchKernelGGL(k_spmv_first<1>, dim3(grid), dim3(256), 0, s, K->nslices, K->nloc, K->d_slot_ptr, K->d_rowof, K->d_cols, K->d_vals, K->d_scale, x, y, partial, stt, k, f, cs);
                else hipLaunchKernelGGL(k_spmv_first<0>, dim3(grid), dim3(256), 0, s, K->nslices, K->nloc, K->d_slot_ptr, K->d_rowof, K->d_cols, K->d_vals, K->d_scale, x, y, partial, stt, k, f, cs);
                mark_scaled(ctx, K);
                return grid;
            }
            return dot == 2 ? launch_spmv_any<2>(ctx, K, kind, x, y, partial, stt, k, which, s, f)
                 : dot == 1 ? launch_spmv_any<1>(ctx, K, kind, x, y, partial, stt, k, which, s, f)
                            : launch_spmv_any<0>(ctx, K, kind, x, y, partial, stt, k, which, s, f);
        };
        unsigned parts = 0;
        bool folded = foldr;
        if (split) {
            HIPCHK(ctx, hipEventRecord(ctx->ev_a, st_));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_a, 0));
            parts = go(1, ctx->side, false);
            HIPCHK(ctx, hipEventRecord(ctx->ev_b, ctx->side));
            STANCHK(halo(x));
            HIPCHK(ctx, hipStreamWaitEvent(st_, ctx->ev_b, 0));
            const unsigned pb = go(2, st_, true);   // adds up the interior launch's partials too
            if (pb == 0) folded = false;            // no boundary slices on this rank: nobody folded
            parts += pb;
        } else {
            if (dist) STANCHK(halo(x));
            parts = go(0, st_, true);
            if (parts == 0) folded = false;
        }
        if (dot && !folded) {
            if (parts > 0 || po.pp) {   // (peer to peer: a rank that owns no rows still sends its zeros)
                const int64_t *sk = k >= 1 ? stt : nullptr;
                if (dot == 2) hipLaunchKernelGGL(k_reduce<2>, dim3(1), dim3(256), 0, st_, partial, (int)parts, out, po, sk, k);
                else hipLaunchKernelGGL(k_reduce<1>, dim3(1), dim3(256), 0, st_, partial, (int)parts, out, po, sk, k);
                n_launch++;
            } else HIPCHK(ctx, hipMemsetAsync(out, 0, 8 * dot, st_));   // a rank that owns no rows
        }
        if (ctx->profiling) hipEventRecord(own ? spmv_ev.back() : spmv64_ev.back(), st_);
        return STAN_OK;
    }

    // v = A^ x and w = A^ x2 in one matrix pass (fused residual refresh), x.v -> out
    int spmv2(double *x, double *x2, double *out, int64_t k, p2p_out po) {
        if (ctx->profiling) {
            hipEvent_t a = events.make(), b = events.make();
            hipEventRecord(a, st_);
            spmv2_ev.push_back(a); spmv2_ev.push_back(b);
            spmv2_k.push_back(k);
        }
        auto go = [&](int which, hipStream_t s, bool last) -> unsigned {
            const fold_args f = last ? fold_to(0, out, po) : NO_FOLD;
            n_launch++;
            if (vs == STAN_PREC_FIXED48) return launch_spmv2<uint32_t>(ctx, K, K->d_vals48, x, x2, v, w, partial, stt, k, which, s, f);
            return vs == STAN_PREC_MIXED ? launch_spmv2<float>(ctx, K, K->d_vals32, x, x2, v, w, partial, stt, k, which, s, f)
                                         : launch_spmv2<double>(ctx, K, K->d_vals, x, x2, v, w, partial, stt, k, which, s, f);
        };
        unsigned parts = 0;
        bool folded = foldr;
        if (split) {
            HIPCHK(ctx, hipEventRecord(ctx->ev_a, st_));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_a, 0));
            parts = go(1, ctx->side, false);
            HIPCHK(ctx, hipEventRecord(ctx->ev_b, ctx->side));
            STANCHK(halo(x));
            STANCHK(halo(x2));
            HIPCHK(ctx, hipStreamWaitEvent(st_, ctx->ev_b, 0));
            const unsigned pb = go(2, st_, true);
            if (pb == 0) folded = false;
            parts += pb;
        } else {
            if (dist) { STANCHK(halo(x)); STANCHK(halo(x2)); }
            parts = go(0, st_, true);
            if (parts == 0) folded = false;
        }
        if (!folded) {
            if (parts > 0 || po.pp) { hipLaunchKernelGGL(k_reduce<1>, dim3(1), dim3(256), 0, st_, partial, (int)parts, out, po, (const int64_t *)stt, k); n_launch++; }
            else HIPCHK(ctx, hipMemsetAsync(out, 0, 8, st_));
        }
        if (ctx->profiling) hipEventRecord(spmv2_ev.back(), st_);
        return STAN_OK;
    }

    int setup();
    int begin_pass(bool from_F, double eps_pass, bool *done);
    int iterate(double eps_pass, int32_t max_its_pass);
    int pass(bool from_F, double eps_pass, int32_t max_its_pass);
    void account_pass();
    int fp64_check(double *xg, const double *b, double *r2_out);
    int finish(const double *x_result, int type, int64_t its, double rel_rec, double rel64, int passes,
               int32_t *term_out, int32_t *iters_out, double *rel_res_out);
};

// ---- set-up ------------------------------------------------------------------------------------------------
int cg_run::setup() {
    st_ = ctx->stream;
    if (ctx->profiling) {
        ev0 = events.make(); ev1 = events.make();
        hipEventRecord(ev0, st_);
    }
    dist = ctx->comm != nullptr || ctx->nranks > 1;  // exchanges in the loop
    // peer to peer (one-process group handle, STAN_OPT_COMM_P2P): no RCCL call below this line
    p2p = dist && ctx->comm_p2p && ctx->p2p != nullptr;
    if (p2p && ctx->p2p->broken.load()) {   // refused at once: not another collective with a peer that is gone
        ctx->err = "cg: the peer-to-peer exchange of this context is broken (a peer rank failed earlier); start a fresh process";
        return STAN_E_COMM;
    }
    // no hipFree while the peers' streams wait for this rank's future exchanges (stan_ctx::defer_frees): peer to peer, and
    // any transport when ranks of this process share the device
    if (p2p || (dist && ctx->peers_share_device)) ctx->defer_frees = true;
    STANCHK(stan_cg_workspace(ctx, K));   // the context's vectors (the placement search probed with them)
    if (p2p) {
        // my neighbours write their boundary rows straight into these vectors: tell them where they are
        const int64_t ns_ = 3 * ((int64_t)K->nslices * 64 + K->nhalo);
        if (!K->d_scale) STANCHK(stan_dmalloc(ctx, &K->d_scale, (size_t)ns_));
        double *const pub[5] = {ctx->ws.xb[0], ctx->ws.xb[1], ctx->ws.p, ctx->ws.r, K->d_scale};
        STANCHK(stan_p2p_publish_vectors(ctx, K, pub));
    }
    // The first product of the loop may scale the matrix on its way (k_spmv_first) instead of a pass of its own: the fp64
    // stream of one rank, the large-system kernel in its default variant, no folded copy (its values are made from the
    // scaled ones), a first product that is a plain one (a residual refresh at iteration 1 is a two-product pass).
    lazy_scale = false;
    if (!K->scaled && ctx->cg_lazy_scaling && !(ctx->comm != nullptr || ctx->nranks > 1) && precision_mode == STAN_PREC_FP64 &&
        !stan_small_system(ctx, K) && ctx->spmv_variant < 0 && ctx->cg_rupdate != 1 && K->nslices > 0) {
        const int rc_plan = stan_matrix_make_folded(ctx, K, STAN_PREC_FP64, true);   // (decides K->fold_state, touches no value)
        if (rc_plan == STAN_E_ALLOC) { stan_matrix_abandon_folding(ctx, K); ctx->err.clear(); }
        else STANCHK(rc_plan);
        lazy_scale = ctx->row_folding == 0 || K->fold_state != 1;
    }
    if (lazy_scale) STANCHK(ensure_scale_vector(ctx, K));
    else STANCHK(ensure_scaled(ctx, K));
    if (ctx->cols16) STANCHK(stan_matrix_make_cols16(ctx, K));
    if (precision_mode == STAN_PREC_MIXED) STANCHK(stan_matrix_make_fp32(ctx, K));
    if (precision_mode == STAN_PREC_FIXED48) STANCHK(stan_matrix_make_fx48(ctx, K));
    // the stream the products really read (FIXED-48 falls back to fp64 when K is not SPD-scalable)
    vs = precision_mode == STAN_PREC_FIXED48 ? (K->d_vals48 ? STAN_PREC_FIXED48 : STAN_PREC_FP64) : precision_mode;
    reduced = precision_mode != STAN_PREC_FP64;
    refine = reduced ? ctx->cg_refine : 0;
    if (!stan_small_system(ctx, K)) {
        const int rc_fold = stan_matrix_make_folded(ctx, K, vs);
        if (rc_fold == STAN_E_ALLOC) {   // an optimisation must not fail the solve: the padded streams serve
            stan_matrix_abandon_folding(ctx, K);
            ctx->err.clear();
        } else
            STANCHK(rc_fold);
    }
    sr = ctx->cg_single_reduce;
    foldr = ctx->cg_fold_reduce;

    n3 = 3 * K->nloc;
    npad = (int64_t)K->nslices * 64;
    ng = 3 * ((npad > K->nloc + K->nhalo ? npad : K->nloc + K->nhalo));
    dof0 = 3 * K->r0;
    xb[0] = ctx->ws.xb[0]; xb[1] = ctx->ws.xb[1]; p = ctx->ws.p; r = ctx->ws.r;
    v = ctx->ws.v; w = ctx->ws.w; bh = ctx->ws.bh;
    if (sr) sv = ctx->ws.sv;
    const unsigned spmv_blocks = stan_small_system(ctx, K) ? (unsigned)K->nslices : nblk(K->nslices, stan_pair_kernel(ctx) ? 2 : 4);
    const size_t npart = 2 * (size_t)(spmv_blocks > VEC_BLOCKS ? spmv_blocks : VEC_BLOCKS) + 16;
    STANCHK(alloc(ctx, bufs, &partial, npart));
    STANCHK(alloc(ctx, bufs, &sc, (size_t)S_NSCAL));
    STANCHK(alloc(ctx, bufs, &stt, (size_t)T_NSTAT));
    STANCHK(alloc(ctx, bufs, &tick, (size_t)(2 * FOLD_WORDS)));   // two ticket-counter sets
    HIPCHK(ctx, hipMemsetAsync(sc, 0, S_NSCAL * 8, st_));
    HIPCHK(ctx, hipMemsetAsync(tick, 0, 2 * FOLD_WORDS * 8, st_));
    HIPCHK(ctx, hipMemsetAsync(xb[0], 0, (size_t)ng * 8, st_));
    HIPCHK(ctx, hipMemsetAsync(xb[1], 0, (size_t)ng * 8, st_));
    HIPCHK(ctx, hipMemsetAsync(p, 0, (size_t)ng * 8, st_));
    if (sr) {
        HIPCHK(ctx, hipMemsetAsync(r, 0, (size_t)ng * 8, st_));
        HIPCHK(ctx, hipMemsetAsync(sv, 0, (size_t)n3 * 8, st_));
    }
    p2p_tab = p2p ? stan_p2p_table(ctx) : nullptr;
    vg = vec_grid(n3);
    its_before_restart = K->n_red > 0 ? K->n_red : 1;   // lincgcreate: ItsBeforeRestart = N (global reduced size)
    // Sharded SpMV with the halo exchange hidden behind the interior slices: the slices whose
    // rows reference no halo column run on a side stream while the main stream packs, sends
    // and receives; the boundary slices follow on the main stream.  RCCL only ever sees the
    // main stream.
    split = dist && ctx->overlap_halo && K->d_sl_bnd != nullptr;
    if (split && !ctx->side) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_a, hipEventDisableTiming));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_b, hipEventDisableTiming));
    }
    h_st = ctx->h_status + SS_H_CG_STATUS;  // pinned
    h_sc = (double *)(ctx->h_status + SS_H_CG_SCALARS);
    poll[0] = events.make(hipEventDisableTiming);
    poll[1] = events.make(hipEventDisableTiming);
    return STAN_OK;
}

// ---- one pass of the loop ------------------------------------------------------------------------------------
// Right-hand side, x0 = 0, r0 = p0 = b^, ||b^||, first residual test.  from_F: b^ = S F (the caller's load vector);
// otherwise b^ = the vector in r (a refinement pass: the fp64 residual fp64_check left there).
int cg_run::begin_pass(bool from_F, double eps_pass, bool *done) {
    {
        fold_args f = fold_to(1, sc + S_VMV, p2p_to(0, true));
        f.nblocks = vg; f.np = (int)vg;
        if (from_F)
            hipLaunchKernelGGL(k_init, dim3(vg), dim3(VEC_T), 0, st_, n3, dof0, K->d_red, d_F, K->d_scale,
                               bh, xb[0], r, p, partial, f);
        else
            hipLaunchKernelGGL(k_init_b, dim3(vg), dim3(VEC_T), 0, st_, n3, (const double *)r, bh, xb[0], r, p, partial, f);
        reduce_if_unfolded((int)vg, 1, sc + S_VMV, p2p_to(0, true));
    }
    red_src rs_b;
    STANCHK(exchange_sums(sc + S_VMV, 1, &rs_b));
    hipLaunchKernelGGL(k_init_scalars, dim3(1), dim3(64), 0, st_, sc, stt, eps_pass, rs_b);
    HIPCHK(ctx, hipGetLastError());
    // status of "iteration 0" (initial residual test)
    HIPCHK(ctx, hipMemcpyAsync(h_st, stt, T_NSTAT * 8, hipMemcpyDeviceToHost, st_));
    STANCHK(cg_wait(ctx, p2p, st_, nullptr));   // (peer to peer: behind the first reduction's wait)
    *done = h_st[T_ITER_A] == 0;
    return STAN_OK;
}

int cg_run::iterate(double eps_pass, int32_t max_its_pass) {
    int64_t k = 1;
    int chunk_id = 0;
    bool done = false;
    int rc = STAN_OK;
    red_src rs_sr{nullptr, 0, nullptr, 0}, rs_vmv{nullptr, 0, nullptr, 0}, rs_r2{nullptr, 0, nullptr, 0};
    // STAN_OPT_CG_REFINE = 2 on a reduced-precision stream: the periodic residual recomputation (alglib's
    // ItsBeforeRUpdate) multiplies with the fp64 values -- the recurrence is re-anchored to the true residual
    // ("reliable updates"), so the literal second product replaces the fused two-product pass
    const bool refresh64 = refine >= 2;
    const int kind_refresh = refresh64 ? STAN_PREC_FP64 : vs;
    // ... and then every REFRESH64 iterations instead of every STAN_OPT_CG_RUPDATE: a refresh on the reduced stream in
    // between would undo the anchoring, and the iteration count does not depend on the period (148^3, fp32 copy: 2346
    // iterations with 10, 20 and 50; profiles/r05/mixed_refine_n148.txt) while every fp64 product costs two of the others
    constexpr int REFRESH64 = 50;
    const int rupdate = refresh64 ? (ctx->cg_rupdate > 0 ? REFRESH64 : 0) : ctx->cg_rupdate;
    if (sr) {   // w_0 = A r_0 with gamma_0, delta_0 (merit_0 = 0 sits in the zeroed scalars)
        if (p2p)         // ... or, peer to peer, is sent as this rank's zero into the slot of the first reduction
            hipLaunchKernelGGL(k_reduce<1>, dim3(1), dim3(256), 0, st_, partial, 0, sc + S_SR_MERIT, p2p_to(2, false), (const int64_t *)nullptr, (int64_t)0);
        rc = spmv(r, w, 2, sc + S_SR_GAMMA, 0, p2p_to(0, true), vs);
        if (rc == STAN_OK) rc = exchange_sums(sc + S_SR_GAMMA, 3, &rs_sr);
    }
    // a sharded loop polls more often: what runs ahead of the stop are exchanges nobody can cut short
    const int chunk = dist ? CHUNK_DIST : CHUNK;
    while (!done && rc == STAN_OK) {
        // enqueue one chunk of iterations
        for (int c = 0; c < chunk && k < hard_cap; c++, k++) {
            const bool refresh = rupdate > 0 && (k % rupdate) == 0;
            if (sr) {
                sr_args a;
                a.n3 = n3; a.k = k; a.sc = sc; a.st = stt; a.epsf = eps_pass; a.maxits = max_its_pass;
                a.its_before_restart = its_before_restart; a.merit_stop = ctx->cg_merit_stop ? 1 : 0;
                a.refresh = refresh ? 1 : 0;
                a.xcur = xb[(k - 1) & 1]; a.xnext = xb[k & 1];
                a.r = r; a.p = p; a.s = sv; a.w = w; a.bh = bh; a.partial = partial;
                a.rs = rs_sr;
                // the merit sum rides in the third column of the coming reduction's mailbox slot (no count of its own)
                a.fold = vec_fold(sc + S_SR_MERIT, p2p_to(2, false));
                hipLaunchKernelGGL(k_vec_sr, dim3(vg), dim3(VEC_T), 0, st_, a);
                n_launch++;
                if (!refresh) reduce_if_unfolded((int)vg, 1, sc + S_SR_MERIT, p2p_to(2, false), k);
                else {   // r' = b^ - A^ x' (ALGLIB's periodic residual recomputation), then as usual
                    rc = spmv(xb[k & 1], v, 0, nullptr, k, NO_P2P, kind_refresh, refresh64);
                    if (rc) break;
                    hipLaunchKernelGGL(k_refresh, dim3(vg), dim3(VEC_T), 0, st_, n3, k, (const int64_t *)stt,
                                       bh, v, xb[k & 1], r, partial, vec_fold(sc + S_SR_DELTA, p2p_to(1, false)));
                    n_launch++;
                    reduce_if_unfolded((int)vg, 2, sc + S_SR_DELTA, p2p_to(1, false), k);   // [r.r (rewritten below), merit]
                }
                rc = spmv(r, w, 2, sc + S_SR_GAMMA, k, p2p_to(0, true), vs);
                if (rc) break;
                rc = exchange_sums(sc + S_SR_GAMMA, 3, &rs_sr);
                if (rc) break;
                continue;
            }
            const bool fused = refresh && ctx->cg_fused_refresh && !refresh64;
            rc = fused ? spmv2(p, xb[(k - 1) & 1], sc + S_VMV, k, p2p_to(0, true))
                       : spmv(p, v, 1, sc + S_VMV, k, p2p_to(0, true), vs);
            if (rc) break;
            rc = exchange_sums(sc + S_VMV, 1, &rs_vmv);
            if (rc) break;
            const p2p_out po_r = p2p_to(0, true);   // the slot of r.r / merit (the wait above moved on to it)
            step_args a;
            a.rs_vmv = rs_vmv;
            a.n3 = n3; a.k = k; a.sc = sc; a.st = stt;
            a.xcur = xb[(k - 1) & 1]; a.xnext = xb[k & 1];
            a.r = r; a.p = p; a.v = v; a.w = w; a.bh = bh; a.partial = partial;
            a.refresh = refresh ? (fused ? 2 : 1) : 0;
            a.merit = ctx->cg_merit_stop ? 1 : 0;
            // x' = x + a p moves into k_update (p is read once for both updates: -79 MB of 714 per
            // iteration at 148^3) unless the merit sum needs x' here or a literal refresh multiplies it
            a.defer_x = (ctx->cg_defer_x && !a.merit && a.refresh != 1) ? 1 : 0;
            a.fold = vec_fold(sc + S_R2NEW, a.refresh == 1 ? NO_P2P : po_r);   // refresh 1: k_refresh forms the sums
            if (ctx->vec_store_nt & 2) hipLaunchKernelGGL(k_step<true>, dim3(vg), dim3(VEC_T), 0, st_, a);
            else hipLaunchKernelGGL(k_step<false>, dim3(vg), dim3(VEC_T), 0, st_, a);
            n_launch++;
            if (a.refresh == 1) {
                // a -5/-4 stop of this iteration is caught by k_refresh/k_update (ITER_B <= k)
                rc = spmv(xb[k & 1], v, 0, nullptr, k, NO_P2P, kind_refresh, refresh64);
                if (rc) break;
                hipLaunchKernelGGL(k_refresh, dim3(vg), dim3(VEC_T), 0, st_, n3, k,
                                   (const int64_t *)stt, bh, v, xb[k & 1], r, partial, vec_fold(sc + S_R2NEW, po_r));
                n_launch++;
            }
            if (!foldr) { reduce_if_unfolded((int)vg, 2, sc + S_R2NEW, po_r, k); n_launch++; }
            rc = exchange_sums(sc + S_R2NEW, 2, &rs_r2);
            if (rc) break;
            const double *ux = a.defer_x ? a.xcur : nullptr;
            double *uxn = a.defer_x ? a.xnext : nullptr;
            if (ctx->vec_store_nt & 1)
                hipLaunchKernelGGL(k_update<true>, dim3(vg), dim3(VEC_T), 0, st_, n3, k, sc, stt, eps_pass,
                                   (int64_t)max_its_pass, its_before_restart, ctx->cg_merit_stop ? 1 : 0, r, p, ux, uxn, rs_r2, rs_vmv);
            else
                hipLaunchKernelGGL(k_update<false>, dim3(vg), dim3(VEC_T), 0, st_, n3, k, sc, stt, eps_pass,
                                   (int64_t)max_its_pass, its_before_restart, ctx->cg_merit_stop ? 1 : 0, r, p, ux, uxn, rs_r2, rs_vmv);
            n_launch++;
        }
        if (rc) break;
        // poll: read the status of the PREVIOUS chunk while this one runs
        int64_t *slot = h_st + 8 * (chunk_id & 1);
        HIPCHK(ctx, hipMemcpyAsync(slot, stt, T_NSTAT * 8, hipMemcpyDeviceToHost, st_));
        HIPCHK(ctx, hipEventRecord(poll[chunk_id & 1], st_));
        if (chunk_id > 0) {
            STANCHK(cg_wait(ctx, p2p, st_, poll[(chunk_id - 1) & 1]));
            int64_t *prev = h_st + 8 * ((chunk_id - 1) & 1);
            if (prev[T_TYPE] != 0) done = true;
        }
        if (k >= hard_cap) done = true;
        if (dist && (ctx->comm_broken.load() || (ctx->p2p && ctx->p2p->broken.load()))) {
            ctx->err = "cg: a peer rank failed (the exchanges were aborted)";
            rc = STAN_E_COMM;
        }
        chunk_id++;
    }
    n_enqueued += k - 1;
    if (p2p) {   // never a blocking wait on a stream that may sit in front of a peer that is gone
        if (rc == STAN_OK) rc = cg_wait(ctx, true, st_, nullptr);
        else stan_p2p_release_own(ctx);
    }
    hipError_t e = hipStreamSynchronize(st_);
    if (rc) return rc;
    if (e != hipSuccess) { ctx->err = std::string("cg: ") + hipGetErrorString(e); return STAN_E_HIP; }
    HIPCHK(ctx, hipMemcpyAsync(h_st, stt, T_NSTAT * 8, hipMemcpyDeviceToHost, st_));
    HIPCHK(ctx, hipMemcpyAsync(h_sc, sc, S_NSCAL * 8, hipMemcpyDeviceToHost, st_));
    HIPCHK(ctx, hipStreamSynchronize(st_));
    pass_type = (int)h_st[T_TYPE];
    pass_its = h_st[T_ITERS];
    if (pass_type == 0) { pass_type = 5; pass_its = k - 1; h_st[T_XSEL] = (k - 1) & 1; }  // hard cap
    xfin = xb[h_st[T_XSEL] & 1];
    return STAN_OK;
}

int cg_run::pass(bool from_F, double eps_pass, int32_t max_its_pass) {
    bool done = false;
    STANCHK(begin_pass(from_F, eps_pass, &done));
    if (done) {   // the first residual test ended it (b = 0, or eps >= 1): no iteration
        HIPCHK(ctx, hipMemcpyAsync(h_sc, sc, S_NSCAL * 8, hipMemcpyDeviceToHost, st_));
        HIPCHK(ctx, hipStreamSynchronize(st_));
        pass_type = (int)h_st[T_TYPE];
        pass_its = h_st[T_ITERS];
        xfin = xb[h_st[T_XSEL] & 1];
        return STAN_OK;
    }
    STANCHK(iterate(eps_pass, max_its_pass));
    account_pass();
    return STAN_OK;
}

// profile: the launch times of the pass that has just ended (its stream is synchronised).  Only launches that did
// work count: the host runs up to two chunks ahead of the status it polls, so a converged solve is followed by a few
// dozen launches that return at once (3-4 us each); averaging those in made the SpMV look ~3 % faster than it is.
void cg_run::account_pass() {
    if (!ctx->profiling) return;
    for (size_t i = 0; i + 1 < spmv_ev.size(); i += 2) {
        if (spmv_k[i / 2] > pass_its) continue;
        float t = 0;
        hipEventElapsedTime(&t, spmv_ev[i], spmv_ev[i + 1]);
        prof_spmv_ms += t;
        prof_spmv_n++;
    }
    for (size_t i = 0; i + 1 < spmv2_ev.size(); i += 2) {
        if (spmv2_k[i / 2] > pass_its) continue;
        float t = 0;
        hipEventElapsedTime(&t, spmv2_ev[i], spmv2_ev[i + 1]);
        prof_spmv2_ms += t;
        prof_spmv2_n++;
    }
    spmv_ev.clear(); spmv_k.clear(); spmv2_ev.clear(); spmv2_k.clear();
}

// ---- the fp64 check of a reduced-precision solve ----------------------------------------------------------------
// r_t = b - A^64 xg with the fp64 values of the scaled matrix (xg: a gather vector -- one of the vectors the peers know,
// cg_run::setup -- holding the iterate on the owned rows); r_t stays in r, *r2_out = ||r_t||^2 over all ranks.
int cg_run::fp64_check(double *xg, const double *b, double *r2_out) {
    STANCHK(spmv(xg, v, 0, nullptr, 0, NO_P2P, STAN_PREC_FP64, true));
    const p2p_out po = p2p_to(0, true);
    hipLaunchKernelGGL(k_refresh, dim3(vg), dim3(VEC_T), 0, st_, n3, (int64_t)0, (const int64_t *)stt, b, v,
                       (const double *)xg, r, partial, vec_fold(sc + S_CHK_R2, po));
    n_launch++;
    reduce_if_unfolded((int)vg, 2, sc + S_CHK_R2, po);
    red_src rs;
    STANCHK(exchange_sums(sc + S_CHK_R2, 2, &rs));
    if (rs.mb) hipLaunchKernelGGL(k_land_sums, dim3(1), dim3(64), 0, st_, sc + S_CHK_R2, rs);   // peer to peer: the mailbox's sums into the local scalars
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(h_sc, sc, S_NSCAL * 8, hipMemcpyDeviceToHost, st_));
    STANCHK(cg_wait(ctx, p2p, st_, nullptr));
    *r2_out = h_sc[S_CHK_R2];
    if (ctx->profiling) {
        for (size_t i = 0; i + 1 < spmv64_ev.size(); i += 2) {
            float t = 0;
            hipEventElapsedTime(&t, spmv64_ev[i], spmv64_ev[i + 1]);
            prof_spmv64_ms += t;
            prof_spmv64_n++;
        }
        spmv64_ev.clear();
    }
    return STAN_OK;
}

// ---- epilogue ------------------------------------------------------------------------------------------------
int cg_run::finish(const double *x_result, int type, int64_t its, double rel_rec, double rel64, int passes,
                   int32_t *term_out, int32_t *iters_out, double *rel_res_out) {
    // U = S x^ on the free DOFs (a rank of a one-process group leaves only ITS entries [u0, u1) of U: the
    // group copies every rank's segment into the caller's buffer, nothing is gathered on the devices)
    if (!dist || ctx->result_segment) {
        hipLaunchKernelGGL(k_result, dim3(vg), dim3(VEC_T), 0, st_, n3, dof0, K->d_red, K->d_scale,
                           x_result, d_U);
    } else {
        double *full;
        STANCHK(alloc(ctx, bufs, &full, (size_t)K->n_dof));
        hipLaunchKernelGGL(k_result_full, dim3(vg), dim3(VEC_T), 0, st_, n3, K->d_scale, x_result,
                           full + dof0);
        STANCHK(stan_comm_allgather_rows(ctx, K, full));
        hipLaunchKernelGGL(k_compress, dim3(vec_grid(K->n_dof)), dim3(VEC_T), 0, st_, K->n_dof,
                           K->d_red, full, d_U);
    }
    HIPCHK(ctx, hipGetLastError());
    if (ctx->profiling) hipEventRecord(ev1, st_);
    HIPCHK(ctx, hipStreamSynchronize(st_));

    if (term_out) *term_out = type;
    if (iters_out) *iters_out = (int32_t)(its > hard_cap ? hard_cap : its);
    if (rel_res_out) *rel_res_out = rel64 >= 0 ? rel64 : rel_rec;
    if (!ctx->profiling) return STAN_OK;
    float ms = 0;
    hipEventElapsedTime(&ms, ev0, ev1);
    stan_profile &pf = ctx->prof;
    pf.cg_ms = ms;
    pf.spmv_ms_total = prof_spmv_ms;
    pf.spmv_launches = prof_spmv_n;
    pf.spmv2_ms_total = prof_spmv2_ms;
    pf.spmv2_launches = prof_spmv2_n;
    pf.fp64_products = (int32_t)prof_spmv64_n;
    pf.fp64_products_ms = prof_spmv64_ms;
    pf.iterations = (int32_t)its;
    pf.termination_type = type;
    pf.rel_residual_recurrence = rel_rec;
    pf.rel_residual_fp64 = rel64;
    pf.refine_passes = passes;
    const int64_t blk_bytes = vs == STAN_PREC_FIXED48 ? 60 : vs == STAN_PREC_MIXED ? 40 : 76;
    // bytes of the format actually streamed: a block of a packed slice carries a 2-B column offset
    // instead of a 4-B index (+ 4 B per slot for its base, shared by 64 rows)
    const bool packed = ctx->cols16 && K->d_cols16;
    const double packed_frac = packed && K->nslots > 0 ? (double)K->slots_packed / (double)K->nslots : 0.0;
    pf.spmv_bytes = K->nblocks * blk_bytes + 3 * K->nloc * 16 + K->nloc * 4
                    - (int64_t)(packed_frac * (double)K->nblocks * 2.0) + (packed ? (K->slots_packed + K->slots_packed2) * 4 : 0);
    const bool folded = ctx->row_folding != 0 && (vs == STAN_PREC_FIXED48 ? K->d_fold_vals48 != nullptr : vs == STAN_PREC_MIXED ? K->d_fold_vals32 != nullptr
                                                                                                                                : K->d_fold_vals != nullptr);
    if (folded) {   // its own packed column stream, 4 B of plan per row
        const bool fp = ctx->cols16 && K->d_fold_cols16;
        const double ff = fp && K->nfslots > 0 ? (double)K->fold_slots_packed / (double)K->nfslots : 0.0;
        pf.spmv_bytes = K->nblocks * blk_bytes + 3 * K->nloc * 16 + K->nloc * 8 - (int64_t)(ff * (double)K->nblocks * 2.0) +
                        (fp ? K->fold_slots_packed * 4 : 0);
        pf.col_slots_packed = fp ? K->fold_slots_packed : 0;
    }
    pf.repacked_streams = folded ? 1 : 0;
    pf.col_slots_packed = packed ? K->slots_packed : 0;
    pf.value_stream = vs;
    // vector passes of one classic iteration: k_step reads r, v (+ p, x unless deferred; + b^ for the
    // merit sum) and writes r (+ x); k_update reads r, p (+ x when deferred) and writes p (+ x)
    pf.cg_iteration_vector_bytes = 3 * K->nloc * 8 * ((ctx->cg_defer_x && !ctx->cg_merit_stop ? 8 : 9) + (ctx->cg_merit_stop ? 1 : 0));
    pf.loop_kernel_launches = n_launch;
    pf.loop_collectives = n_coll;
    pf.loop_stream_waits = n_wait;
    pf.loop_iterations_enqueued = n_enqueued;
    // what the stream spent in the exchanges (RCCL launches, or peer-to-peer waits): events around each
    auto sum_pairs = [](const std::vector<hipEvent_t> &ev, double *tot, int64_t *cnt) {
        *tot = 0; *cnt = 0;
        for (size_t i = 0; i + 1 < ev.size(); i += 2) {
            float t = 0;
            if (hipEventElapsedTime(&t, ev[i], ev[i + 1]) == hipSuccess) { *tot += t; (*cnt)++; }
        }
    };
    sum_pairs(red_ev, &pf.comm_reduce_ms_total, &pf.comm_reduce_calls);
    sum_pairs(halo_ev, &pf.comm_halo_ms_total, &pf.comm_halo_calls);
    return STAN_OK;
}

}  // namespace

int stan_cg_device(stan_ctx *ctx, stan_matrix *K, const double *d_F, double eps_f,
                   int32_t max_its, int32_t precision_mode, double *d_U, int32_t *term_out,
                   int32_t *iters_out, double *rel_res_out) {
    if (precision_mode != STAN_PREC_FP64 && precision_mode != STAN_PREC_MIXED &&
        precision_mode != STAN_PREC_FIXED48) {
        ctx->err = "cg_solve: unknown precision_mode";
        return STAN_E_UNSUPPORTED;
    }
    if (eps_f < 0 || max_its < 0) {
        ctx->err = "cg_solve: eps_f and max_its must be >= 0";
        return STAN_E_ARG;
    }
    if (eps_f == 0 && max_its == 0) eps_f = 1.0e-6;  // lincgsetcond
    // no hipFree while the peers' streams wait for this rank's future exchanges (stan_ctx::defer_frees); the guard
    // outlives the run object, whose buffers are released into the deferred list
    struct free_later { stan_ctx *c; ~free_later() { if (c->defer_frees) { stan_flush_deferred(c); stan_p2p_ipc_trim(c); } } } free_guard{ctx};
    cg_run R{ctx, K, d_F, eps_f, max_its, precision_mode, d_U};
    STANCHK(R.setup());

    constexpr int MAX_PASSES = 8;
    int type = 0, passes = 0;
    int64_t its = 0;
    double bnorm0 = 0, rel_rec = 0, rel64 = -1.0, prev_rel64 = 0, eps_pass = eps_f;
    const double *x_result = nullptr;
    for (;;) {
        const int64_t left = max_its > 0 ? (int64_t)max_its - its : 0;
        STANCHK(R.pass(passes == 0, eps_pass, (int32_t)left));
        passes++;
        its += R.pass_its;
        type = R.pass_type;
        if (passes == 1) bnorm0 = R.h_sc[S_BNORM];
        // ||r|| / ||b|| of the loop's own recurrence, against the ORIGINAL right-hand side (what alglib reports)
        rel_rec = bnorm0 > 0 ? std::sqrt(R.h_sc[S_R2OUT]) / bnorm0 : 0.0;
        x_result = R.xfin;
        if (!R.reduced) break;
        // ---- a reduced-precision mode: what was delivered, in fp64 ----
        double *xg = const_cast<double *>(R.xfin);
        if (passes > 1) {   // the passes' iterates add up: x = x_1 + d_2 + ...; gathered from p (a vector the peers know)
            hipLaunchKernelGGL(k_accumulate, dim3(R.vg), dim3(VEC_T), 0, R.st_, R.n3, R.xacc, R.xfin, R.p);
            xg = R.p;
            x_result = R.xacc;
        }
        double r2 = 0;
        STANCHK(R.fp64_check(xg, passes > 1 ? R.b0 : R.bh, &r2));
        rel64 = bnorm0 > 0 ? std::sqrt(r2) / bnorm0 : 0.0;
        const bool met = rel64 <= eps_f || bnorm0 == 0;
        if (type != 1) break;                       // 5, 7, -4, -5: reported as the loop ended, with the fp64 residual
        if (met) break;
        // the recurrence says converged, the fp64 residual does not: another pass on r_t (iterative refinement) ...
        const bool out_of_its = max_its > 0 && its >= max_its;
        const bool stalled = passes > 1 && !(rel64 < 0.5 * prev_rel64);
        if (R.refine == 0 || passes >= MAX_PASSES || out_of_its || stalled || !std::isfinite(rel64)) {
            type = out_of_its ? 5 : 7;             // ... or the truth: no further progress at this precision (alglib's 7)
            break;
        }
        if (passes == 1) {
            STANCHK(alloc(ctx, R.bufs, &R.xacc, (size_t)(R.n3 > 0 ? R.n3 : 1)));
            STANCHK(alloc(ctx, R.bufs, &R.b0, (size_t)(R.n3 > 0 ? R.n3 : 1)));
            HIPCHK(ctx, hipMemcpyAsync(R.xacc, R.xfin, (size_t)R.n3 * 8, hipMemcpyDeviceToDevice, R.st_));
            HIPCHK(ctx, hipMemcpyAsync(R.b0, R.bh, (size_t)R.n3 * 8, hipMemcpyDeviceToDevice, R.st_));
        }
        prev_rel64 = rel64;
        eps_pass = eps_f / rel64;                   // ||r|| <= eps ||b0|| with ||b_pass|| = ||r_t|| = rel64 ||b0||
        HIPCHK(ctx, hipMemsetAsync(R.sc, 0, S_NSCAL * 8, R.st_));
        if (R.sr) HIPCHK(ctx, hipMemsetAsync(R.sv, 0, (size_t)R.n3 * 8, R.st_));
    }
    return R.finish(x_result, type, its, rel_rec, rel64, passes, term_out, iters_out, rel_res_out);
}

// y = K x on the reduced system (test helper; single rank)
int stan_spmv_reduced(stan_ctx *ctx, stan_matrix *K, const double *d_x, double *d_y) {
    if (ctx->nranks != 1) { ctx->err = "spmv: single-rank contexts only"; return STAN_E_UNSUPPORTED; }
    hipStream_t st_ = ctx->stream;
    const int64_t n3 = 3 * K->nloc, npad3 = 3 * (int64_t)K->nslices * 64;
    dev_bufs bufs;
    double *xf, *yf; int64_t *stt;
    STANCHK(alloc(ctx, bufs, &xf, (size_t)npad3));
    STANCHK(alloc(ctx, bufs, &yf, (size_t)npad3));
    STANCHK(alloc(ctx, bufs, &stt, (size_t)T_NSTAT));
    int64_t init[T_NSTAT] = {0x7fffffffffffffffLL, 0x7fffffffffffffffLL, 0, 0, 0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(stt, init, sizeof(init), hipMemcpyHostToDevice, st_));
    HIPCHK(ctx, hipMemsetAsync(xf, 0, (size_t)npad3 * 8, st_));
    const double *sdiv = K->scaled ? K->d_scale : nullptr;
    // K x = S^-1 (A^ (S^-1 x)) when the matrix already carries its scaling
    hipLaunchKernelGGL(k_expand, dim3(vec_grid(n3)), dim3(VEC_T), 0, st_, n3, (int64_t)0, K->d_red,
                       d_x, sdiv, xf);
    launch_spmv<double, 0>(ctx, K, K->d_vals, xf, yf, nullptr, stt, 1);
    // compress (and undo the row scaling)
    hipLaunchKernelGGL(k_compress_div, dim3(vec_grid(n3)), dim3(VEC_T), 0, st_, n3, K->d_red, sdiv, yf, d_y);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(st_));
    return STAN_OK;
}

// diag[d - red[d]] = K_dd on the free DOFs (single rank)
int stan_matrix_diagonal(stan_ctx *ctx, stan_matrix *K, double *d_diag) {
    if (ctx->nranks != 1) { ctx->err = "matrix_diagonal: single-rank contexts only"; return STAN_E_UNSUPPORTED; }
    hipStream_t st_ = ctx->stream;
    const int64_t n3 = 3 * K->nloc;
    dev_bufs bufs;
    double *full;
    STANCHK(alloc(ctx, bufs, &full, (size_t)(n3 > 0 ? n3 : 1)));
    if (K->nloc > 0)
        hipLaunchKernelGGL(k_diag_get, dim3(nblk(K->nloc, 256)), dim3(256), 0, st_, K->nloc, K->d_rowlen, K->d_posof,
                           K->d_slot_ptr, K->d_cols, K->d_vals, K->scaled ? K->d_scale : (const double *)nullptr, full);
    hipLaunchKernelGGL(k_compress_div, dim3(vec_grid(n3)), dim3(VEC_T), 0, st_, n3, K->d_red, (const double *)nullptr, full, d_diag);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(st_));
    return STAN_OK;
}

// y_owned = A_local x_local, x_local = [owned rows | halo columns] (plan checks; any rank)
int stan_spmv_local(stan_ctx *ctx, stan_matrix *K, const double *d_x, double *d_y) {
    dev_bufs bufs;
    int64_t *stt;
    STANCHK(alloc(ctx, bufs, &stt, (size_t)T_NSTAT));
    int64_t init[T_NSTAT] = {0x7fffffffffffffffLL, 0x7fffffffffffffffLL, 0, 0, 0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(stt, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    if (K->d_sl_bnd) {  // sharded: interior + boundary lists must cover every slice exactly once
        HIPCHK(ctx, hipMemsetAsync(d_y, 0xff, (size_t)(3 * K->nloc) * 8, ctx->stream));  // NaN
        launch_spmv<double, 0>(ctx, K, K->d_vals, d_x, d_y, nullptr, stt, 1, 1);
        launch_spmv<double, 0>(ctx, K, K->d_vals, d_x, d_y, nullptr, stt, 1, 2);
    } else
        launch_spmv<double, 0>(ctx, K, K->d_vals, d_x, d_y, nullptr, stt, 1);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return STAN_OK;
}

int stan_spmv_bench_device(stan_ctx *ctx, stan_matrix *K, int32_t precision_mode, int32_t reps,
                           double *avg_ms) {
    hipStream_t st_ = ctx->stream;
    const bool mixed = precision_mode == STAN_PREC_MIXED;
    if (mixed) STANCHK(stan_matrix_make_fp32(ctx, K));
    if (precision_mode == STAN_PREC_FIXED48) {
        STANCHK(ensure_scaled(ctx, K));
        STANCHK(stan_matrix_make_fx48(ctx, K));
        if (!K->d_vals48) { ctx->err = "spmv_bench: matrix not representable in FIXED48"; return STAN_E_UNSUPPORTED; }
    }
    const bool fx = precision_mode == STAN_PREC_FIXED48;
    const int64_t npad = (int64_t)K->nslices * 64;
    const int64_t ng = 3 * ((npad > K->nloc + K->nhalo ? npad : K->nloc + K->nhalo));
    dev_bufs bufs;
    double *x, *y, *partial; int64_t *stt;
    // the CG's own gather vector and product buffer: the pair (value block, vector blocks) that is
    // timed here is the pair the solve will run on
    STANCHK(stan_cg_workspace(ctx, K));
    x = ctx->ws.p; y = ctx->ws.v;
    STANCHK(alloc(ctx, bufs, &partial, 2 * (size_t)K->nslices + 2));   // k_spmv_small leaves one partial per slice
    STANCHK(alloc(ctx, bufs, &stt, (size_t)T_NSTAT));
    int64_t init[T_NSTAT] = {0x7fffffffffffffffLL, 0x7fffffffffffffffLL, 0, 0, 0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(stt, init, sizeof(init), hipMemcpyHostToDevice, st_));
    hipLaunchKernelGGL(k_fill, dim3(vec_grid(ng)), dim3(VEC_T), 0, st_, x, ng, 1.0);
    event_bag events;
    hipEvent_t a = events.make(), b = events.make();
    auto one = [&]() {
        if (mixed) launch_spmv<float, 1>(ctx, K, K->d_vals32, x, y, partial, stt, 1);
        else if (fx) launch_spmv<uint32_t, 1>(ctx, K, K->d_vals48, x, y, partial, stt, 1);
        else launch_spmv<double, 1>(ctx, K, K->d_vals, x, y, partial, stt, 1);
    };
    for (int i = 0; i < 3; i++) one();
    hipEventRecord(a, st_);
    for (int i = 0; i < reps; i++) one();
    hipEventRecord(b, st_);
    HIPCHK(ctx, hipEventSynchronize(b));
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    *avg_ms = reps > 0 ? ms / reps : 0;
    HIPCHK(ctx, hipGetLastError());
    return STAN_OK;
}

// `reps` sweeps of k_value_stream over K's resident fp64 values (see the kernel): average ms per sweep and the bytes one
// sweep reads (the slots' values: padded slots are streamed like real ones, as the product streams them).
int stan_stream_bench_device(stan_ctx *ctx, stan_matrix *K, int32_t reps, double *avg_ms, int64_t *bytes) {
    hipStream_t st_ = ctx->stream;
    *avg_ms = 0;
    *bytes = (int64_t)K->nslots * 64 * 72;
    if (K->nslices <= 0 || !K->d_vals) return STAN_OK;
    const unsigned grid = nblk(K->nslices, 4);
    dev_bufs bufs;
    double *sink;
    STANCHK(alloc(ctx, bufs, &sink, (size_t)grid));
    event_bag events;
    hipEvent_t a = events.make(), b = events.make();
    auto one = [&]() { hipLaunchKernelGGL(k_value_stream, dim3(grid), dim3(256), 0, st_, K->nslices, K->d_slot_ptr, K->d_vals, sink); };
    for (int i = 0; i < 3; i++) one();
    hipEventRecord(a, st_);
    for (int i = 0; i < reps; i++) one();
    hipEventRecord(b, st_);
    HIPCHK(ctx, hipEventSynchronize(b));
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    *avg_ms = ms / reps;
    HIPCHK(ctx, hipGetLastError());
    return STAN_OK;
}

// Time of the fp64 SpMV of K streaming its values from `vals` (any contents: only the addresses
// matter), median of 3 launches after a warm-up.  Used by the allocation-by-trial of placement.hip.
// self_pair: the gather vector and the product are carved out of the FRONT of the candidate block
// itself instead of the context's vectors -- by construction the same-group (slow) pairing, i.e.
// the reference the search compares the real pairing with (profiles/r02/placement_cross_self_n148.txt).
int stan_spmv_probe(stan_ctx *ctx, stan_matrix *K, const void *vals, size_t bytes, int32_t precision,
                    float *ms_out, bool self_pair) {
    hipStream_t st_ = ctx->stream;
    *ms_out = 0;
    if (K->nslices <= 0) return STAN_OK;
    const int64_t npad = (int64_t)K->nslices * 64;
    const int64_t ng = 3 * ((npad > K->nloc + K->nhalo ? npad : K->nloc + K->nhalo));
    dev_bufs bufs;
    double *x, *y, *partial; int64_t *stt;
    // the CG's own gather vector and product buffer: the pair (value block, vector blocks) that is
    // timed here is the pair the solve will run on
    STANCHK(stan_cg_workspace(ctx, K));
    x = ctx->ws.p; y = ctx->ws.v;
    if (self_pair) {   // the block holds no values yet (only addresses matter to the timing)
        // the gather vector and the product must fit into the candidate: a stream with few slots per
        // slice (or a large halo) has no self-paired reference -- *ms_out stays 0, the search then
        // keeps the fastest real pairing (placement.hip)
        if ((size_t)(((ng + 511) & ~(int64_t)511) + 3 * K->nloc) * 8 > bytes) return STAN_OK;
        x = (double *)const_cast<void *>(vals);
        y = x + ((ng + 511) & ~(int64_t)511);
    }
    STANCHK(alloc(ctx, bufs, &partial, 2 * (size_t)K->nslices + 2));   // k_spmv_small leaves one partial per slice
    STANCHK(alloc(ctx, bufs, &stt, (size_t)T_NSTAT));
    int64_t init[T_NSTAT] = {0x7fffffffffffffffLL, 0x7fffffffffffffffLL, 0, 0, 0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(stt, init, sizeof(init), hipMemcpyHostToDevice, st_));
    hipLaunchKernelGGL(k_fill, dim3(vec_grid(ng)), dim3(VEC_T), 0, st_, x, ng, 1.0);
    event_bag ev;
    // one launch to warm up, then two groups of three launches back to back, the faster group counts.  (Round 4: single
    // launches between host synchronisations -- the first form of this probe -- start on an idle device and read 2-3 %
    // under the same product inside a sequence of kernels.)
    auto one = [&]() {
        if (precision == STAN_PREC_FIXED48)
            launch_spmv<uint32_t, 1>(ctx, K, (const uint32_t *)vals, x, y, partial, stt, 1);
        else if (precision == STAN_PREC_MIXED)
            launch_spmv<float, 1>(ctx, K, (const float *)vals, x, y, partial, stt, 1);
        else
            launch_spmv<double, 1>(ctx, K, (const double *)vals, x, y, partial, stt, 1);
    };
    one();
    float best = 0;
    for (int g = 0; g < 2; g++) {
        hipEvent_t a = ev.make(), b = ev.make();
        hipEventRecord(a, st_);
        for (int r = 0; r < 3; r++) one();
        hipEventRecord(b, st_);
        HIPCHK(ctx, hipEventSynchronize(b));
        float t = 0;
        hipEventElapsedTime(&t, a, b);
        if (g == 0 || t < best) best = t;
    }
    HIPCHK(ctx, hipGetLastError());
    *ms_out = best / 3;
    return STAN_OK;
}


// the matrix carries S K S from its first solve on -- or from its first export: stan_hip_matrix_to_csr divides the
// scaled values on the way out, whether or not a solve has happened, so an export before a solve and one after it are
// the same bits (rounds 1-4 un-scaled the values in place for an export and re-scaled them for the next solve)
int stan_matrix_ensure_scaled(stan_ctx *ctx, stan_matrix *K) { return ensure_scaled(ctx, K); }
