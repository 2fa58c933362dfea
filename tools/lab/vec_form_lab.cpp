// vec_form_lab.cpp -- the shape of the CG's vector kernels (lab, not product)
// hipcc -O3 --offload-arch=gfx950 -o /tmp/vec_form_lab tools/lab/vec_form_lab.cpp && /tmp/vec_form_lab
// k_step's traffic (r' = r - a v, r'.r' per block; 237 MB) behind a cold 6.4 GB sweep, as it is in the CG, in several shapes:
// grid-stride loops over 2048 ... 16384 blocks (the library: 2048), two / four independent elements per trip, one element per
// thread over n / 256 blocks, 16-B accesses.  And k_update's traffic (p' = r + b p, x' = x + a p: three reads, two writes).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ double blk_sum(double v, double *sh) {
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

template <bool NTL> __device__ __forceinline__ double ldv(const double *p) { if (NTL) return __builtin_nontemporal_load(p); return *p; }

template <int U, bool NTL = true>   // grid-stride, U independent elements per trip
__global__ void __launch_bounds__(256) k_step_gs(long long n, double *r, const double *__restrict__ v, double a, double *partial) {
    __shared__ double sh[4];
    const long long stride = (long long)gridDim.x * 256;
    double s = 0;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        double x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; u++) { x[u] = ldv<NTL>(r + i + u * stride); y[u] = ldv<NTL>(v + i + u * stride); }
#pragma unroll
        for (int u = 0; u < U; u++) { const double c = x[u] - a * y[u]; __builtin_nontemporal_store(c, r + i + u * stride); s += c * c; }
    }
    for (; i < n; i += stride) { const double c = ldv<NTL>(r + i) - a * ldv<NTL>(v + i); __builtin_nontemporal_store(c, r + i); s += c * c; }
    const double t = blk_sum(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ void __launch_bounds__(256) k_step_one(long long n, double *r, const double *__restrict__ v, double a, double *partial) {
    __shared__ double sh[4];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    double s = 0;
    if (i < n) { const double c = __builtin_nontemporal_load(r + i) - a * __builtin_nontemporal_load(v + i); __builtin_nontemporal_store(c, r + i); s = c * c; }
    const double t = blk_sum(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

typedef double d2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_step_v2(long long n2, d2 *r, const d2 *__restrict__ v, double a, double *partial) {
    __shared__ double sh[4];
    const long long stride = (long long)gridDim.x * 256;
    double s = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) {
        const d2 x = __builtin_nontemporal_load(r + i), y = __builtin_nontemporal_load(v + i);
        d2 c; c.x = x.x - a * y.x; c.y = x.y - a * y.y;
        __builtin_nontemporal_store(c, r + i);
        s += c.x * c.x + c.y * c.y;
    }
    const double t = blk_sum(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

template <int U, bool NTL = true>
__global__ void __launch_bounds__(256) k_update_gs(long long n, const double *__restrict__ r, double *p, const double *__restrict__ xc, double *xn,
                                                   double a, double b) {
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        double rr[U], pp[U], xx[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            rr[u] = ldv<NTL>(r + i + u * stride); pp[u] = ldv<NTL>(p + i + u * stride);
            xx[u] = ldv<NTL>(xc + i + u * stride);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            __builtin_nontemporal_store(xx[u] + a * pp[u], xn + i + u * stride);
            __builtin_nontemporal_store(rr[u] + b * pp[u], p + i + u * stride);
        }
    }
    for (; i < n; i += stride) {
        const double pi = ldv<NTL>(p + i);
        __builtin_nontemporal_store(ldv<NTL>(xc + i) + a * pi, xn + i);
        __builtin_nontemporal_store(ldv<NTL>(r + i) + b * pi, p + i);
    }
}

__global__ void __launch_bounds__(256) k_sweep(const double *__restrict__ p, long long n, double *sink) {
    const long long stride = (long long)gridDim.x * 256;
    double a = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) a += __builtin_nontemporal_load(p + i);
    if (a == 0.1234567890123) sink[0] = a;
}

template <class F>
static float cold(F launch, const double *big, double *sink, int reps) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float sum = 0;
    for (int i = 0; i < reps + 2; i++) {
        hipLaunchKernelGGL(k_sweep, dim3(256 * 24), dim3(256), 0, 0, big, (long long)800 << 20, sink);
        (void)hipEventRecord(a, 0);
        launch();
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        if (i >= 2) sum += ms;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return sum / reps * 1e3f;
}

int main() {
    const long long n = 9857244, pad = (n + 511) & ~511LL;
    double *vec, *big, *partial, *sink;
    CK(hipMalloc(&vec, (size_t)pad * 8 * 6));
    CK(hipMalloc(&partial, 65536 * 8));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&big, (size_t)800 << 23));
    CK(hipMemset(vec, 0, (size_t)pad * 8 * 6));
    CK(hipMemset(big, 0, (size_t)800 << 23));
    double *r = vec, *v = vec + pad, *p = vec + 2 * pad, *xc = vec + 3 * pad, *xn = vec + 4 * pad;
    printf("k_step's traffic (237 MB), cold, us per launch\n");
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        printf("  grid-stride, %5d blocks: 1 per trip %.2f | 2 per trip %.2f | 4 per trip %.2f | 16-B accesses %.2f\n", blocks,
               cold([&] { hipLaunchKernelGGL(k_step_gs<1>, dim3(blocks), dim3(256), 0, 0, n, r, v, 0.5, partial); }, big, sink, 20),
               cold([&] { hipLaunchKernelGGL(k_step_gs<2>, dim3(blocks), dim3(256), 0, 0, n, r, v, 0.5, partial); }, big, sink, 20),
               cold([&] { hipLaunchKernelGGL(k_step_gs<4>, dim3(blocks), dim3(256), 0, 0, n, r, v, 0.5, partial); }, big, sink, 20),
               cold([&] { hipLaunchKernelGGL(k_step_v2, dim3(blocks), dim3(256), 0, 0, n / 2, (d2 *)r, (const d2 *)v, 0.5, partial); }, big, sink, 20));
    }
    for (int blocks : {2048, 8192})
        printf("  PLAIN loads, %5d blocks: 1 per trip %.2f | 2 per trip %.2f\n", blocks,
               cold([&] { hipLaunchKernelGGL((k_step_gs<1, false>), dim3(blocks), dim3(256), 0, 0, n, r, v, 0.5, partial); }, big, sink, 20),
               cold([&] { hipLaunchKernelGGL((k_step_gs<2, false>), dim3(blocks), dim3(256), 0, 0, n, r, v, 0.5, partial); }, big, sink, 20));
    printf("  one element per thread, %lld blocks: %.2f\n", (n + 255) / 256,
           cold([&] { hipLaunchKernelGGL(k_step_one, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, r, v, 0.5, partial); }, big, sink, 20));
    printf("k_update's traffic (394 MB), cold, us per launch\n");
    for (int blocks : {1024, 2048, 4096, 8192, 16384})
        printf("  grid-stride, %5d blocks: 1 per trip %.2f | 2 per trip %.2f | 4 per trip %.2f\n", blocks,
               cold([&] { hipLaunchKernelGGL(k_update_gs<1>, dim3(blocks), dim3(256), 0, 0, n, r, p, xc, xn, 0.5, 0.25); }, big, sink, 20),
               cold([&] { hipLaunchKernelGGL(k_update_gs<2>, dim3(blocks), dim3(256), 0, 0, n, r, p, xc, xn, 0.5, 0.25); }, big, sink, 20),
               cold([&] { hipLaunchKernelGGL(k_update_gs<4>, dim3(blocks), dim3(256), 0, 0, n, r, p, xc, xn, 0.5, 0.25); }, big, sink, 20));
    for (int blocks : {2048, 8192})
        printf("  PLAIN loads, %5d blocks: 1 per trip %.2f | 2 per trip %.2f\n", blocks,
               cold([&] { hipLaunchKernelGGL((k_update_gs<1, false>), dim3(blocks), dim3(256), 0, 0, n, r, p, xc, xn, 0.5, 0.25); }, big, sink, 20),
               cold([&] { hipLaunchKernelGGL((k_update_gs<2, false>), dim3(blocks), dim3(256), 0, 0, n, r, p, xc, xn, 0.5, 0.25); }, big, sink, 20));
    return 0;
}
