// csr_lab.hip -- measurement aid, not on the product path: the same operator in scalar CSR
// (fp64 values + int32 columns, what SURVEY.md section 8d prices the SpMV in and what alglib's
// CRS holds), built on the device from the BSELL-64 matrix, and a CSR-vector SpMV over it.
// It exists to put a number next to the format decision: BSELL-64 moves 76 B per 3x3 block,
// CSR 108 B.  stan_hip_csr_spmv_bench() times it with HIP events like stan_hip_spmv_bench().
#include <algorithm>
#include <cmath>

#include "../internal.h"
#include "stan_hip_lab.h"

namespace {

__global__ void k_csr_rowlen(int64_t nloc, const int32_t *rowlen, int32_t *len3) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d < 3 * nloc) len3[d] = 3 * rowlen[d / 3];
}

__global__ void k_csr_fill(int64_t nloc, const int32_t *rowlen, const int32_t *posof, const int32_t *slot_ptr,
                           const int32_t *cols, const double *vals, const int64_t *rp, int32_t *ci,
                           double *cv) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nloc) return;
    const int64_t slice = posof[row] >> 6;
    const int lane = (int)(posof[row] & 63);
    const int32_t k0 = slot_ptr[slice];
    for (int k = 0; k < rowlen[row]; k++) {
        const int32_t c = cols[((int64_t)k0 + k) * 64 + lane];
        const double *v = vals + ((int64_t)k0 + k) * 9 * 64 + lane;
        for (int m = 0; m < 3; m++)
            for (int n = 0; n < 3; n++) {
                const int64_t q = rp[3 * row + m] + 3 * k + n;
                ci[q] = 3 * c + n;
                cv[q] = v[(3 * m + n) * 64];
            }
    }
}

// CSR-vector: 32 lanes per scalar row (rows hold ~81 entries), 8 rows per 256-thread block
__global__ void __launch_bounds__(256)
k_csr_spmv(int64_t nrows, const int64_t *__restrict__ rp, const int32_t *__restrict__ ci,
           const double *__restrict__ cv, const double *__restrict__ x, double *__restrict__ y) {
    const int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int l = threadIdx.x & 31;
    double a = 0;
    if (row < nrows) {
        const int64_t q0 = rp[row], q1 = rp[row + 1];
        for (int64_t q = q0 + l; q < q1; q += 32)
            a += __builtin_nontemporal_load(cv + q) * x[__builtin_nontemporal_load(ci + q)];
    }
#pragma unroll
    for (int d = 16; d > 0; d >>= 1) a += __shfl_xor(a, d, 32);
    if (row < nrows && l == 0) y[row] = a;
}

inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

}  // namespace

extern "C" int stan_hip_csr_spmv_bench(stan_ctx *ctx, stan_matrix *K, int32_t reps, double *avg_ms,
                                       int64_t *bytes_per_launch, double *max_rel_diff) {
    if (!ctx || !K || !avg_ms || reps <= 0 || K->ctx != ctx) return STAN_E_ARG;
    if (ctx->nranks != 1) { ctx->err = "csr_spmv_bench: single-rank contexts only"; return STAN_E_UNSUPPORTED; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t nloc = K->nloc, n3 = 3 * nloc, nnz = 9 * K->nblocks;
    int32_t *len3 = nullptr, *ci = nullptr;
    int64_t *rp = nullptr;
    double *cv = nullptr, *x = nullptr, *y = nullptr, *y2 = nullptr;
    int64_t *stt = nullptr;
    std::vector<void *> own;
    auto A = [&](auto **p, size_t n) { int rc = stan_dmalloc(ctx, p, n); if (!rc) own.push_back((void *)*p); return rc; };
    struct F { stan_ctx *c; std::vector<void *> &v; ~F() { for (void *q : v) stan_dfree(c, q); } } fr{ctx, own};
    const int64_t npad3 = 3 * (int64_t)K->nslices * 64;
    STANCHK(A(&len3, (size_t)n3 + 1)); STANCHK(A(&rp, (size_t)n3 + 2)); STANCHK(A(&ci, (size_t)nnz));
    STANCHK(A(&cv, (size_t)nnz)); STANCHK(A(&x, (size_t)npad3)); STANCHK(A(&y, (size_t)npad3));
    STANCHK(A(&y2, (size_t)npad3)); STANCHK(A(&stt, 8));
    hipLaunchKernelGGL(k_csr_rowlen, dim3(nblk(n3, 256)), dim3(256), 0, st, nloc, K->d_rowlen, len3);
    STANCHK(stan_scan_exclusive(ctx, len3, rp, n3));
    hipLaunchKernelGGL(k_csr_fill, dim3(nblk(nloc, 256)), dim3(256), 0, st, nloc, K->d_rowlen, K->d_posof, K->d_slot_ptr,
                       K->d_cols, K->d_vals, rp, ci, cv);
    // x_i = 1 + (i mod 7): cheap, non-constant
    std::vector<double> hx((size_t)npad3);
    for (size_t i = 0; i < hx.size(); i++) hx[i] = 1.0 + (double)(i % 7);
    HIPCHK(ctx, hipMemcpyAsync(x, hx.data(), hx.size() * 8, hipMemcpyHostToDevice, st));
    hipEvent_t a, b;
    event_bag events;
    a = events.make(); b = events.make();
    for (int i = 0; i < 3; i++)
        hipLaunchKernelGGL(k_csr_spmv, dim3(nblk(n3, 8)), dim3(256), 0, st, n3, rp, ci, cv, x, y);
    hipEventRecord(a, st);
    for (int i = 0; i < reps; i++)
        hipLaunchKernelGGL(k_csr_spmv, dim3(nblk(n3, 8)), dim3(256), 0, st, n3, rp, ci, cv, x, y);
    hipEventRecord(b, st);
    HIPCHK(ctx, hipEventSynchronize(b));
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    *avg_ms = ms / reps;
    if (bytes_per_launch) *bytes_per_launch = nnz * 12 + n3 * 8 /* rowptr (int64 here) */ + n3 * 16;
    // agreement with the BSELL-64 product on the same vector
    if (max_rel_diff) {
        STANCHK(stan_spmv_local(ctx, K, x, y2));
        std::vector<double> h1((size_t)n3), h2((size_t)n3);
        HIPCHK(ctx, hipMemcpy(h1.data(), y, (size_t)n3 * 8, hipMemcpyDeviceToHost));
        HIPCHK(ctx, hipMemcpy(h2.data(), y2, (size_t)n3 * 8, hipMemcpyDeviceToHost));
        double mx = 0, df = 0;
        for (int64_t i = 0; i < n3; i++) {
            mx = std::max(mx, std::fabs(h2[(size_t)i]));
            df = std::max(df, std::fabs(h1[(size_t)i] - h2[(size_t)i]));
        }
        *max_rel_diff = mx > 0 ? df / mx : 0;
    }
    HIPCHK(ctx, hipGetLastError());
    return STAN_OK;
}
