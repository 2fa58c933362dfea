// fold.hip -- folded rows: a re-packed copy of the BSELL-64 streams in which the long rows of a slice lend
// their tails to the idle slots of its short rows, so that a wave walks ~(blocks of the slice)/64 slots instead
// of (blocks of its longest row).
//
// Why: the reference reads arbitrary CHEXA meshes (Database.cs:39-111).  On a mesh whose rows differ in length
// the padded layout makes a wave execute the slots of its slice's longest row, and the SpMV's time follows that
// count -- not the bytes (a build whose streams carried no padding: -15 % HBM traffic, -2 % time) and not the gathers
// (profiles/r03/SELL_C_SIGMA.md, addendum).  Sorting rows of equal length into one slice (SELL-C-sigma) cuts the
// count but takes 64 consecutive breadth-first rows apart, which costs more in gather locality than it saves.
// Folding cuts it WITHOUT moving a row out of its slice:
//   * W = the smallest width for which the plan below works (>= ceil(blocks of the slice / 64));
//   * a row longer than W keeps its first W blocks in its own lane; the rest goes, in order, into the free slots
//     [len_j, W) of short lanes j, taken from the last lane downwards; a lane serves ONE foreign row;
//   * the kernel (cg.hip: k_spmv_fold) adds the products of slots < own[lane] to the lane's own row and the
//     others to a second accumulator; after the loop the second accumulators go through LDS and every owner adds
//     its helpers' partial sums in a fixed order (descending lane).
// A folded row is summed as own part + piece + piece ...: another order than the padded layout's, fixed by the
// layout (deterministic, the same for a shard and the whole matrix: shards are cut on slice boundaries).  Rows
// that are not folded keep their bits.
#include "internal.h"

namespace {

// one wavefront per slice; lane 0 plans (<= 64 rows: a few hundred scalar steps), every lane stores its entry
__global__ void __launch_bounds__(256)
k_fold_plan(int32_t nslices, int64_t nloc, const int32_t *slot_ptr, const int32_t *rowof, const int32_t *rowlen,
            int32_t *width, int4 *plan, uint32_t *meta, unsigned long long *unsorted) {
    __shared__ int32_t s_len[4][64], s_owner[4][64], s_off[4][64], s_take[4][64], s_hfirst[4][64], s_nh[4][64];
    __shared__ int32_t s_w[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t slice = (int64_t)blockIdx.x * 4 + w;
    if (slice < nslices) {
        const int64_t row = rowof[slice * 64 + lane];
        s_len[w][lane] = row < nloc ? rowlen[row] : 0;
    }
    __syncthreads();
    if (slice < nslices && lane == 0) {
        const int32_t *l = s_len[w];
        int32_t T = 0;
        bool sorted = true;
        for (int i = 0; i < 64; i++) { T += l[i]; sorted &= i == 0 || l[i] <= l[i - 1]; }
        const int32_t lmax = slot_ptr[slice + 1] - slot_ptr[slice];
        int32_t W = sorted ? (T + 63) / 64 : lmax;   // unsorted slices (a matrix of another build) are copied as they are
        if (!sorted) atomicAdd(unsorted, 1ULL);
        for (;; W++) {
            for (int i = 0; i < 64; i++) { s_owner[w][i] = -1; s_off[w][i] = 0; s_take[w][i] = 0; s_hfirst[w][i] = 0; s_nh[w][i] = 0; }
            if (W >= lmax) { W = lmax; break; }
            int j = 63;
            bool ok = true;
            for (int i = 0; i < 64 && l[i] > W && ok; i++) {
                int32_t e = l[i] - W, off = W;
                s_hfirst[w][i] = j;
                while (e > 0) {
                    if (j <= i || l[j] >= W || s_nh[w][i] >= 255) { ok = false; break; }
                    const int32_t room = W - l[j], take = e < room ? e : room;
                    s_owner[w][j] = i; s_off[w][j] = off; s_take[w][j] = take;
                    off += take; e -= take; s_nh[w][i]++; j--;
                }
            }
            if (ok) break;
        }
        s_w[w] = W;
        width[slice] = W;
    }
    __syncthreads();
    if (slice < nslices) {
        const int32_t W = s_w[w], own = s_len[w][lane] < W ? s_len[w][lane] : W;
        plan[slice * 64 + lane] = make_int4(own, s_owner[w][lane], s_off[w][lane], s_take[w][lane]);
        meta[slice * 64 + lane] = (uint32_t)own | ((uint32_t)s_hfirst[w][lane] << 16) | ((uint32_t)s_nh[w][lane] << 24);
    }
}

// folded slot k of lane j <- padded slot of the plan (own row, or the owner's row at its offset), or a zero entry
template <typename VT, int ROWS>
__global__ void __launch_bounds__(256)
k_fold_fill(int32_t nslices, const int32_t *slot_ptr, const int32_t *fold_ptr, const int4 *plan, const int32_t *cols,
            const VT *in, int32_t *ocols, VT *out) {
    const int lane = threadIdx.x & 63;
    const int64_t slice = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slice >= nslices) return;
    const int4 p = plan[slice * 64 + lane];   // own slots, owner lane, offset in the owner's row, slots taken
    const int64_t k0 = slot_ptr[slice], f0 = fold_ptr[slice];
    const int32_t W = fold_ptr[slice + 1] - fold_ptr[slice], lmax = slot_ptr[slice + 1] - slot_ptr[slice];
    for (int32_t k = 0; k < W; k++) {
        int64_t src = -1;   // padded (slot * 64 + lane) of the source entry
        if (k < p.x) src = (k0 + k) * 64 + lane;
        else if (p.y >= 0 && k - p.x < p.w) src = (k0 + p.z + (k - p.x)) * 64 + p.y;
        const int64_t dst = (f0 + k) * 64 + lane;
        if (ocols) ocols[dst] = src >= 0 ? cols[src] : (lmax > 0 ? cols[k0 * 64 + lane] : 0);   // a padding entry points at a valid column
#pragma unroll
        for (int j = 0; j < ROWS; j++) {
            VT v;
            if (src >= 0) v = in[(src / 64 * ROWS + j) * 64 + (src & 63)];
            else if (sizeof(VT) == 4 && ROWS == 14) v = (VT)(j < 9 ? 0u : 0x80008000u);   // FIXED-48: offset-binary zero
            else v = (VT)0;
            out[((f0 + k) * ROWS + j) * 64 + lane] = v;
        }
    }
}

}  // namespace

void stan_matrix_drop_folded_values(stan_ctx *ctx, stan_matrix *K) {
    if (K->d_fold_vals) { stan_dfree(ctx, K->d_fold_vals); K->d_fold_vals = nullptr; }
    if (K->d_fold_vals32) { stan_dfree(ctx, K->d_fold_vals32); K->d_fold_vals32 = nullptr; }
    if (K->d_fold_vals48) { stan_dfree(ctx, K->d_fold_vals48); K->d_fold_vals48 = nullptr; }
}

// no memory for a second copy of the streams: give everything back, the padded streams serve
void stan_matrix_abandon_folding(stan_ctx *ctx, stan_matrix *K) {
    stan_matrix_drop_folded_values(ctx, K);
    for (void **q : {(void **)&K->d_fold_ptr, (void **)&K->d_fold_meta, (void **)&K->d_fold_plan, (void **)&K->d_fold_cols,
                     (void **)&K->d_fold_cols16, (void **)&K->d_fold_colbase, (void **)&K->d_fold_pair_ptr, (void **)&K->d_fold_packed}) {
        stan_dfree(ctx, *q);
        *q = nullptr;
    }
    K->fold_cols_filled = false;
    K->fold_state = -1;
}

// Folded copies of the column stream and of the value stream `stream_kind`, built when STAN_OPT_ROW_FOLDING asks
// for them (1: always; -1, the default: when the plan saves more than 5 % of the slots).  The padded streams stay: scaling,
// export and the placement search work on them.
int stan_matrix_make_folded(stan_ctx *ctx, stan_matrix *K, int32_t stream_kind, bool plan_only) {
    // -2 = declined by the AUTO threshold only: STAN_OPT_ROW_FOLDING = 1 ("always") set afterwards examines the matrix again
    if (K->fold_state == -2 && ctx->row_folding == 1) K->fold_state = 0;
    if (ctx->row_folding == 0 || K->fold_state < 0 || K->nslots <= 0 || K->nslices <= 0) return STAN_OK;
    hipStream_t st = ctx->stream;
    const unsigned grid = (unsigned)((K->nslices + 3) / 4);
    if (K->fold_state == 0) {
        int32_t *width = nullptr;
        int64_t *ptr64 = nullptr;
        struct tmp { stan_ctx *c; int32_t **a; int64_t **b; ~tmp() { stan_dfree(c, *a); stan_dfree(c, *b); } } guard{ctx, &width, &ptr64};
        // (an earlier attempt that ran out of memory may have left these behind)
        stan_dfree(ctx, K->d_fold_plan); K->d_fold_plan = nullptr;
        stan_dfree(ctx, K->d_fold_meta); K->d_fold_meta = nullptr;
        STANCHK(stan_dmalloc(ctx, &width, (size_t)K->nslices + 1));
        STANCHK(stan_dmalloc(ctx, &ptr64, (size_t)K->nslices + 2));
        STANCHK(stan_dmalloc(ctx, &K->d_fold_plan, (size_t)K->nslices * 64));
        STANCHK(stan_dmalloc(ctx, &K->d_fold_meta, (size_t)K->nslices * 64));
        unsigned long long *d_uns = (unsigned long long *)(ctx->d_status + SS_COUNTER);
        HIPCHK(ctx, hipMemsetAsync(d_uns, 0, 8, st));
        hipLaunchKernelGGL(k_fold_plan, dim3(grid), dim3(256), 0, st, K->nslices, K->nloc, K->d_slot_ptr, K->d_rowof, K->d_rowlen,
                           width, K->d_fold_plan, K->d_fold_meta, d_uns);
        STANCHK(stan_scan_exclusive(ctx, width, ptr64, K->nslices));
        std::vector<int64_t> h((size_t)K->nslices + 1);
        HIPCHK(ctx, hipMemcpyAsync(h.data(), ptr64, h.size() * 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        K->nfslots = h[(size_t)K->nslices];
        // not worth a second copy of the matrix: a layout that is already tight (the cube: 0.7 % padding)
        if (K->nfslots >= ((int64_t)1 << 31) || (ctx->row_folding < 0 && (double)K->nfslots > 0.95 * (double)K->nslots)) {
            stan_dfree(ctx, K->d_fold_plan); K->d_fold_plan = nullptr;
            stan_dfree(ctx, K->d_fold_meta); K->d_fold_meta = nullptr;
            K->fold_state = K->nfslots >= ((int64_t)1 << 31) ? -1 : -2;
            return STAN_OK;
        }
        std::vector<int32_t> h32(h.size());
        for (size_t i = 0; i < h.size(); i++) h32[i] = (int32_t)h[i];
        STANCHK(stan_dmalloc(ctx, &K->d_fold_ptr, h32.size()));
        HIPCHK(ctx, hipMemcpyAsync(K->d_fold_ptr, h32.data(), h32.size() * 4, hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipStreamSynchronize(st));   // h32 must outlive the copy
        STANCHK(stan_dmalloc(ctx, &K->d_fold_cols, (size_t)(K->nfslots > 0 ? K->nfslots : 1) * 64));
        K->fold_state = 1;
    }
    if (plan_only) return STAN_OK;   // the caller only wants K->fold_state decided (cg.hip: may the first product scale the matrix?)
    const size_t n = (size_t)(K->nfslots > 0 ? K->nfslots : 1) * 64;
    if (!K->fold_cols_filled) {
        // the columns first (k_fold_fill without values), then their packed stream (16-bit offsets from a per-slot
        // base, cg.hip colstream): the placement search below times the real product on its candidates
        hipLaunchKernelGGL((k_fold_fill<double, 0>), dim3(grid), dim3(256), 0, st, K->nslices, K->d_slot_ptr, K->d_fold_ptr,
                           K->d_fold_plan, K->d_cols, (const double *)nullptr, K->d_fold_cols, (double *)nullptr);
        HIPCHK(ctx, hipGetLastError());
        K->fold_cols_filled = true;
        if (ctx->cols16)
            STANCHK(stan_pack_columns(ctx, K->nslices, K->nfslots, K->d_fold_ptr, K->d_fold_cols, &K->d_fold_cols16, &K->d_fold_colbase,
                                      &K->d_fold_pair_ptr, &K->d_fold_packed, &K->fold_slots_packed));
    }
    // the value block is allocated by trial like the padded one (placement.hip): the folded product is timed on
    // every candidate (stan_ctx::fold_probe tells the dispatch that the candidate is a FOLDED stream)
    auto streamed = [&](void **p, size_t bytes, int32_t prec) {
        return stan_dmalloc_streamed(ctx, p, bytes, [&, bytes, prec](const void *q, float *ms, bool self) {
            ctx->fold_probe = q;
            const int rc = stan_spmv_probe(ctx, K, q, bytes, prec, ms, self);
            ctx->fold_probe = nullptr;
            return rc;
        });
    };
    if (stream_kind == STAN_PREC_MIXED) {
        if (!K->d_fold_vals32 && K->d_vals32) {
            STANCHK(streamed((void **)&K->d_fold_vals32, n * 9 * 4, STAN_PREC_MIXED));
            hipLaunchKernelGGL((k_fold_fill<float, 9>), dim3(grid), dim3(256), 0, st, K->nslices, K->d_slot_ptr, K->d_fold_ptr,
                               K->d_fold_plan, K->d_cols, K->d_vals32, (int32_t *)nullptr, K->d_fold_vals32);
        }
    } else if (stream_kind == STAN_PREC_FIXED48) {
        if (!K->d_fold_vals48 && K->d_vals48) {
            STANCHK(streamed((void **)&K->d_fold_vals48, n * 14 * 4, STAN_PREC_FIXED48));
            hipLaunchKernelGGL((k_fold_fill<uint32_t, 14>), dim3(grid), dim3(256), 0, st, K->nslices, K->d_slot_ptr, K->d_fold_ptr,
                               K->d_fold_plan, K->d_cols, K->d_vals48, (int32_t *)nullptr, K->d_fold_vals48);
        }
    } else if (!K->d_fold_vals) {
        STANCHK(streamed((void **)&K->d_fold_vals, n * 9 * 8, STAN_PREC_FP64));
        hipLaunchKernelGGL((k_fold_fill<double, 9>), dim3(grid), dim3(256), 0, st, K->nslices, K->d_slot_ptr, K->d_fold_ptr,
                           K->d_fold_plan, K->d_cols, K->d_vals, (int32_t *)nullptr, K->d_fold_vals);
    }
    HIPCHK(ctx, hipGetLastError());
    return STAN_OK;
}
