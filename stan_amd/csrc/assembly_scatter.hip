// assembly_scatter.hip -- the north-star numeric assembly, kept as a measured ALTERNATIVE
// (STAN_OPT_ASSEMBLY_MODE = 1): one element per wavefront computes K_e = sum_g B'DB |J| w with
// the nodal coordinates, the Gauss-point data and the 24x24 K_e staged in LDS, then a
// colour-ordered, atomic-free scatter adds its 64 3x3 blocks into the BSELL-64 values
// (elements of one colour share no node, colours run one after the other in a fixed order, so
// the sum is deterministic).  Element colours come from a parallel greedy colouring with hashed
// priorities (Jones-Plassmann style) over the node->element incidence lists.
// The default path is the row-owner gather of assembly.hip; DESIGN.md has the comparison.
// Replaces the same reference code: SolverFunctions.cs:129-174 (Parallel.ForEach + lock(K) +
// sparseadd), Element.cs:118-155.
#include <vector>

#include "internal.h"
#include "hex8_device.h"

namespace {

__device__ __forceinline__ unsigned prio(int32_t e) {
    unsigned x = (unsigned)e * 2654435761u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    return x;
}

// tentative colour = smallest colour no already-coloured neighbour uses.  Colours are examined in windows of 64 (one
// mask word): a structured mesh needs 8-18, the elements around a high-valence node (72 collapsed hexes on the axis
// of a revolved mesh) one each -- the window moves on until a free colour turns up, so there is no limit.
__global__ void k_col_tentative(int64_t n_elem, const int32_t *conn, const int32_t *perm,
                                const int64_t *ptr, const int32_t *list, const int32_t *colour,
                                int32_t *tent) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_elem || colour[e] >= 0) return;
    for (int32_t base = 0;; base += 64) {
        unsigned long long forbid = 0;
        for (int a = 0; a < 8; a++) {
            const int64_t row = perm[conn[e * 8 + a]];
            for (int64_t q = ptr[row]; q < ptr[row + 1]; q++) {
                const int32_t e2 = list[q] >> 3;
                const int32_t c2 = colour[e2];
                if (e2 != e && c2 >= base && c2 < base + 64) forbid |= 1ull << (c2 - base);
            }
        }
        if (~forbid != 0ull) { tent[e] = base + __ffsll((long long)~forbid) - 1; return; }
    }
}

// keep the tentative colour unless an uncoloured neighbour with higher priority wants it too
__global__ void k_col_resolve(int64_t n_elem, const int32_t *conn, const int32_t *perm,
                              const int64_t *ptr, const int32_t *list, const int32_t *colour,
                              const int32_t *tent, int32_t *colour_out, unsigned long long *remaining) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_elem) return;
    if (colour[e] >= 0) { colour_out[e] = colour[e]; return; }
    const int32_t c = tent[e];
    const unsigned pe = prio((int32_t)e);
    bool lose = false;
    for (int a = 0; a < 8 && !lose; a++) {
        const int64_t row = perm[conn[e * 8 + a]];
        for (int64_t q = ptr[row]; q < ptr[row + 1]; q++) {
            const int32_t e2 = list[q] >> 3;
            if (e2 == e || colour[e2] >= 0 || tent[e2] != c) continue;
            const unsigned p2 = prio(e2);
            if (p2 > pe || (p2 == pe && e2 > e)) { lose = true; break; }
        }
    }
    colour_out[e] = lose ? -1 : c;
    if (lose) atomicAdd(remaining, 1ull);
}

__global__ void k_col_max(int64_t n_elem, const int32_t *colour, int32_t *out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int v = e < n_elem ? colour[e] : -1;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    if ((threadIdx.x & 63) == 0 && v >= 0) atomicMax(out, v);
}
__global__ void k_col_count(int64_t n_elem, const int32_t *colour, int32_t *cnt) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_elem) atomicAdd(&cnt[colour[e]], 1);
}
__global__ void k_col_fill(int64_t n_elem, const int32_t *colour, const int32_t *off, int32_t *cursor,
                           int32_t *order) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_elem) order[off[colour[e]] + atomicAdd(&cursor[colour[e]], 1)] = (int32_t)e;
}

// one wavefront per element of the current colour: K_e in LDS, then 64 block read-modify-writes
__global__ void __launch_bounds__(256)
k_scatter(int32_t count, const int32_t *order, const int32_t *conn, const int32_t *perm,
          const double *xyz, const int32_t *elem_mat, const uint8_t *elem_type, const double *mat_lamG,
          const int32_t *rowlen, const int32_t *posof, const int32_t *slot_ptr, const int32_t *cols, double *vals,
          long long *bad_elem) {
    __shared__ double xs[4][24];
    __shared__ double gp[4][80];
    __shared__ double ke[4][576];  // the 24x24 element stiffness, row-major
    __shared__ int32_t ids[4][8];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int32_t idx = blockIdx.x * 4 + w;
    const bool valid = idx < count;
    const int32_t e = valid ? order[idx] : 0;
    const int a = lane >> 3, b = lane & 7;
    int32_t na = 0, nb = 0, type = STAN_HEX8_G2;
    if (valid) {
        na = conn[(int64_t)e * 8 + a];
        nb = conn[(int64_t)e * 8 + b];
        type = elem_type[e];
        if (lane < 24) xs[w][lane] = xyz[3 * (int64_t)conn[(int64_t)e * 8 + lane / 3] + lane % 3];
        if (lane < 8) ids[w][lane] = conn[(int64_t)e * 8 + lane];
    }
    __syncthreads();
    if (valid && lane < 8) {
        double o[10];
        const double det = hex8_gp_setup(xs[w], type, lane, o);
        if (det == 0.0 && hex8_gauss_weight(type, lane) != 0.0) atomicMin(bad_elem, (long long)e);
#pragma unroll
        for (int j = 0; j < 10; j++) gp[w][lane * 10 + j] = o[j];
    }
    __syncthreads();
    if (valid) {
        const int32_t m = elem_mat[e];
        double kb[9];
        hex8_block_ab(gp[w], 10, type, a, b, mat_lamG[2 * m], mat_lamG[2 * m + 1], kb);
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) ke[w][(3 * a + r) * 24 + 3 * b + c] = kb[3 * r + c];
    }
    __syncthreads();
    if (!valid) return;
    // scatter block (a, b): row = DOF block of node a, column = DOF block of node b
    const int64_t row = perm[na];
    const int32_t col = perm[nb];
    const int64_t slice = posof[row] >> 6;   // SELL-C-sigma: where the row sits in the sliced layout
    const int rl = (int)(posof[row] & 63);
    const int32_t k0 = slot_ptr[slice];
    int lo = 0, hi = rowlen[row] - 1, pos = -1;
    while (lo <= hi) {  // columns ascend (single rank: local == global)
        const int mid = (lo + hi) >> 1;
        const int32_t c = cols[((int64_t)k0 + mid) * 64 + rl];
        if (c == col) { pos = mid; break; }
        if (c < col) lo = mid + 1; else hi = mid - 1;
    }
    if (pos < 0) return;
    // a degenerate element lists a node twice: its duplicate (a,b) pairs hit one block; let the
    // first pair add the sum of all of them
    double add[9];
#pragma unroll
    for (int j = 0; j < 9; j++) add[j] = 0.0;
    bool first = true;
    for (int a2 = 0; a2 < 8; a2++)
        for (int b2 = 0; b2 < 8; b2++) {
            const bool same = ids[w][a2] == na && ids[w][b2] == nb;
            if (!same) continue;
            if (a2 * 8 + b2 < lane) first = false;
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int c = 0; c < 3; c++) add[3 * r + c] += ke[w][(3 * a2 + r) * 24 + 3 * b2 + c];
        }
    if (!first) return;
    double *v = vals + ((int64_t)k0 + pos) * 9 * 64 + rl;
#pragma unroll
    for (int j = 0; j < 9; j++) v[j * 64] += add[j];
}

// essential BCs on the assembled values: fixed rows/columns zero, fixed diagonal one
__global__ void __launch_bounds__(256)
k_apply_bc(int32_t nslices, int64_t nloc, const int32_t *slot_ptr, const int32_t *rowof, const int32_t *rowlen,
           const int32_t *cols, double *vals, const uint8_t *fixmask) {
    const int lane = threadIdx.x & 63;
    const int64_t slice = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slice >= nslices) return;
    const int64_t row = rowof[slice * 64 + lane];
    if (row >= nloc) return;
    const int rf = fixmask[row];
    const int32_t k0 = slot_ptr[slice];
    for (int k = 0; k < rowlen[row]; k++) {
        const int32_t c = cols[((int64_t)k0 + k) * 64 + lane];
        const int cf = fixmask[c];
        if (!rf && !cf) continue;
        double *v = vals + ((int64_t)k0 + k) * 9 * 64 + lane;
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int n = 0; n < 3; n++)
                if (((rf >> m) & 1) || ((cf >> n) & 1)) v[(3 * m + n) * 64] = (c == row && m == n) ? 1.0 : 0.0;
    }
}

inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

}  // namespace

int stan_assemble_colour_scatter(stan_ctx *ctx, stan_matrix *K, int64_t n_elem, const int32_t *d_conn,
                                 const int32_t *d_perm, const double *d_xyz, const int32_t *d_elem_mat,
                                 const uint8_t *d_elem_type, const double *d_lamG, const int64_t *d_ptr,
                                 const int32_t *d_list, long long *d_bad) {
    hipStream_t st = ctx->stream;
    HIPCHK(ctx, hipMemsetAsync(K->d_vals, 0, (size_t)K->nslots * 9 * 64 * 8, st));
    ctx->prof_colours = 0;
    if (n_elem > 0) {
        int32_t *d_col[2], *d_tent, *d_cnt = nullptr, *d_order, *d_maxc;
        unsigned long long *d_rem;
        std::vector<void *> owned;
        auto A = [&](auto **p, size_t n) { int rc = stan_dmalloc(ctx, p, n); if (!rc) owned.push_back((void *)*p); return rc; };
        struct F { stan_ctx *c; std::vector<void *> &v; ~F() { for (void *q : v) stan_dfree(c, q); } } fr{ctx, owned};
        STANCHK(A(&d_col[0], (size_t)n_elem)); STANCHK(A(&d_col[1], (size_t)n_elem));
        STANCHK(A(&d_tent, (size_t)n_elem)); STANCHK(A(&d_maxc, 2)); STANCHK(A(&d_order, (size_t)n_elem));
        STANCHK(A(&d_rem, 1));
        HIPCHK(ctx, hipMemsetAsync(d_col[0], 0xff, (size_t)n_elem * 4, st));  // -1 = uncoloured
        int cur = 0;
        for (int round = 0; round < 4096; round++) {
            HIPCHK(ctx, hipMemsetAsync(d_rem, 0, 8, st));
            hipLaunchKernelGGL(k_col_tentative, dim3(nblk(n_elem, 256)), dim3(256), 0, st, n_elem, d_conn,
                               d_perm, d_ptr, d_list, d_col[cur], d_tent);
            hipLaunchKernelGGL(k_col_resolve, dim3(nblk(n_elem, 256)), dim3(256), 0, st, n_elem, d_conn,
                               d_perm, d_ptr, d_list, d_col[cur], d_tent, d_col[cur ^ 1], d_rem);
            cur ^= 1;
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_status + SS_COUNTER, d_rem, 8, hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            if (ctx->h_status[SS_COUNTER] == 0) break;
        }
        if (ctx->h_status[SS_COUNTER] != 0) { ctx->err = "element colouring did not converge"; return STAN_E_HIP; }
        // elements grouped by colour (as many colours as the mesh needed)
        HIPCHK(ctx, hipMemsetAsync(d_maxc, 0xff, 4, st));
        hipLaunchKernelGGL(k_col_max, dim3(nblk(n_elem, 256)), dim3(256), 0, st, n_elem, d_col[cur], d_maxc);
        int32_t maxc = -1;
        HIPCHK(ctx, hipMemcpyAsync(&maxc, d_maxc, 4, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        const int32_t nc = maxc + 1;
        STANCHK(A(&d_cnt, (size_t)3 * nc));
        HIPCHK(ctx, hipMemsetAsync(d_cnt, 0, (size_t)3 * nc * 4, st));
        hipLaunchKernelGGL(k_col_count, dim3(nblk(n_elem, 256)), dim3(256), 0, st, n_elem, d_col[cur], d_cnt);
        std::vector<int32_t> cnt((size_t)nc), off((size_t)nc);
        HIPCHK(ctx, hipMemcpyAsync(cnt.data(), d_cnt, (size_t)nc * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        int32_t run = 0, ncol = 0;
        for (int c = 0; c < nc; c++) { off[(size_t)c] = run; run += cnt[(size_t)c]; if (cnt[(size_t)c]) ncol = c + 1; }
        HIPCHK(ctx, hipMemcpyAsync(d_cnt + nc, off.data(), (size_t)nc * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_col_fill, dim3(nblk(n_elem, 256)), dim3(256), 0, st, n_elem, d_col[cur],
                           d_cnt + nc, d_cnt + 2 * nc, d_order);
        HIPCHK(ctx, hipStreamSynchronize(st));   // off[] must outlive the copy
        ctx->prof_colours = ncol;
        // colour passes, in colour order
        for (int c = 0; c < ncol; c++)
            if (cnt[c] > 0)
                hipLaunchKernelGGL(k_scatter, dim3(nblk(cnt[c], 4)), dim3(256), 0, st, cnt[c], d_order + off[c],
                                   d_conn, d_perm, d_xyz, d_elem_mat, d_elem_type, d_lamG, K->d_rowlen,
                                   K->d_posof, K->d_slot_ptr, K->d_cols, K->d_vals, d_bad);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(st));  // temporaries are freed on return
    }
    if (K->nslices > 0)
        hipLaunchKernelGGL(k_apply_bc, dim3(nblk(K->nslices, 4)), dim3(256), 0, st, K->nslices, K->nloc,
                           K->d_slot_ptr, K->d_rowof, K->d_rowlen, K->d_cols, K->d_vals, K->d_fixmask);
    HIPCHK(ctx, hipGetLastError());
    return STAN_OK;
}
