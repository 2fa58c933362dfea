// scan.hip -- exclusive prefix sum used by the symbolic phase (row pointers of the
// node->element incidence lists, ELL slot pointers, halo compaction).
// Integer work, HBM-bound: each element is read twice and written once.
#include "internal.h"

namespace {

constexpr int SCAN_T = 256;          // threads per block
constexpr int SCAN_ITEMS = 8;        // items per thread
constexpr int SCAN_TILE = SCAN_T * SCAN_ITEMS;

__device__ inline int64_t wave_incl_scan(int64_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int64_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// block-wide exclusive scan of one value per thread; returns exclusive prefix, *total = sum
__device__ inline int64_t block_excl_scan(int64_t v, int64_t *total) {
    __shared__ int64_t wsum[SCAN_T / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int64_t inc = wave_incl_scan(v);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int64_t off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_T / 64; i++) {
        if (i < w) off += wsum[i];
        tot += wsum[i];
    }
    __syncthreads();
    *total = tot;
    return off + inc - v;
}

template <typename Tin>
__global__ void __launch_bounds__(SCAN_T) k_tile_sums(const Tin *in, int64_t *tile_sum, int64_t n) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    int64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        int64_t idx = base + (int64_t)i * SCAN_T + threadIdx.x;
        if (idx < n) s += (int64_t)in[idx];
    }
    int64_t tot;
    block_excl_scan(s, &tot);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = tot;
}

template <typename Tin>
__global__ void __launch_bounds__(SCAN_T) k_tile_scan(const Tin *in, const int64_t *tile_off,
                                                      int64_t *out, int64_t n) {
    // thread t owns the contiguous items [t*ITEMS, t*ITEMS+ITEMS) of the tile
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t v[SCAN_ITEMS];
    int64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        int64_t idx = base + i;
        v[i] = idx < n ? (int64_t)in[idx] : 0;
        s += v[i];
    }
    int64_t tot;
    int64_t off = block_excl_scan(s, &tot) + (tile_off ? tile_off[blockIdx.x] : 0);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        int64_t idx = base + i;
        if (idx < n) out[idx] = off;
        off += v[i];
    }
}

__global__ void k_set_total(const int64_t *excl_last, const int32_t *in_last32,
                            const int64_t *in_last64, int64_t *out_total) {
    *out_total = *excl_last + (in_last32 ? (int64_t)*in_last32 : *in_last64);
}

template <typename Tin>
int scan_rec(stan_ctx *ctx, const Tin *d_in, int64_t *d_out, int64_t n) {
    // out[0..n-1] exclusive, out[n] total
    if (n <= 0) {
        HIPCHK(ctx, hipMemsetAsync(d_out, 0, sizeof(int64_t), ctx->stream));
        return STAN_OK;
    }
    const int64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (ntiles == 1) {
        hipLaunchKernelGGL(k_tile_scan<Tin>, dim3(1), dim3(SCAN_T), 0, ctx->stream, d_in,
                           (const int64_t *)nullptr, d_out, n);
    } else {
        int64_t *d_tsum = nullptr, *d_toff = nullptr;
        STANCHK(stan_dmalloc(ctx, &d_tsum, (size_t)ntiles));
        STANCHK(stan_dmalloc(ctx, &d_toff, (size_t)ntiles + 1));
        hipLaunchKernelGGL(k_tile_sums<Tin>, dim3((unsigned)ntiles), dim3(SCAN_T), 0, ctx->stream,
                           d_in, d_tsum, n);
        int rc = scan_rec<int64_t>(ctx, d_tsum, d_toff, ntiles);
        if (rc == STAN_OK)
            hipLaunchKernelGGL(k_tile_scan<Tin>, dim3((unsigned)ntiles), dim3(SCAN_T), 0,
                               ctx->stream, d_in, (const int64_t *)d_toff, d_out, n);
        // pooled blocks are reused in stream order; smaller ones go through hipFree, which waits
        // for the device itself
        stan_dfree(ctx, d_tsum);
        stan_dfree(ctx, d_toff);
        STANCHK(rc);
    }
    if (sizeof(Tin) == 4)
        hipLaunchKernelGGL(k_set_total, dim3(1), dim3(1), 0, ctx->stream, d_out + (n - 1),
                           (const int32_t *)d_in + (n - 1), (const int64_t *)nullptr, d_out + n);
    else
        hipLaunchKernelGGL(k_set_total, dim3(1), dim3(1), 0, ctx->stream, d_out + (n - 1),
                           (const int32_t *)nullptr, (const int64_t *)d_in + (n - 1), d_out + n);
    HIPCHK(ctx, hipGetLastError());
    return STAN_OK;
}

}  // namespace

int stan_scan_exclusive(stan_ctx *ctx, const int32_t *d_in, int64_t *d_out, int64_t n) {
    return scan_rec<int32_t>(ctx, d_in, d_out, n);
}
