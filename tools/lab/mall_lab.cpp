// mall_lab.cpp -- what does a RE-READ cost on this device?  (lab, not product)
// hipcc -O3 --offload-arch=gfx950 -o /tmp/mall_lab tools/lab/mall_lab.cpp && /tmp/mall_lab
// Question behind it: a symmetric-storage SpMV would stream the upper blocks of K once (3.4 GB at 148^3) and read every block
// a second time, transposed, from the row of its column: a re-read that trails the stream by the matrix' bandwidth (tens of
// MB).  If the memory-side cache (256 MB) serves that re-read, HBM traffic halves.
//   (1) the same S bytes read again and again, S = 8 MB ... 2 GB: where the rate drops is what the caches hold;
//   (2) a 3.2 GB stream where every wave also reads the run that lies LAG bytes behind its own: time against the plain stream
//       and against streaming 2 x the bytes; first reads non-temporal or plain.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

constexpr long long RUN = 15552;   // doubles per wave, the SpMV's run length (124 KB)

template <int NT>
__device__ __forceinline__ double ld(const double *p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

template <int NT1, int NT2>
__global__ void __launch_bounds__(256) k_stream(const double *__restrict__ p, long long n, long long lag, double *sink) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long b0 = wave * RUN;
    if (b0 >= n) return;
    const long long b1 = b0 + RUN < n ? b0 + RUN : n;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    long long i = b0 + lane;
    if (lag == 0) {
        for (; i + 192 < b1; i += 256) {
            a0 += ld<NT1>(p + i);
            a1 += ld<NT1>(p + i + 64);
            a2 += ld<NT1>(p + i + 128);
            a3 += ld<NT1>(p + i + 192);
        }
    } else {
        const long long back = b0 >= lag ? lag : 0;   // the first LAG bytes have nothing behind them: read themselves
        for (; i + 192 < b1; i += 256) {
            a0 += ld<NT1>(p + i);
            a1 += ld<NT1>(p + i + 64);
            a2 += ld<NT1>(p + i + 128);
            a3 += ld<NT1>(p + i + 192);
            a0 += ld<NT2>(p + i - back);
            a1 += ld<NT2>(p + i + 64 - back);
            a2 += ld<NT2>(p + i + 128 - back);
            a3 += ld<NT2>(p + i + 192 - back);
        }
    }
    const double s = (a0 + a1) + (a2 + a3);
    if (s == 0.1234567890123) sink[0] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NT1, int NT2>
static float run(const double *p, long long n, long long lag, double *sink, int reps) {
    const long long waves = (n + RUN - 1) / RUN;
    const int grid = (int)((waves + 3) / 4);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k_stream<NT1, NT2>), dim3(grid), dim3(256), 0, 0, p, n, lag, sink);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_stream<NT1, NT2>), dim3(grid), dim3(256), 0, 0, p, n, lag, sink);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms / reps;
}

int main() {
    const long long NBIG = 400LL << 20;   // doubles: 3.2 GB
    double *p, *sink;
    CK(hipMalloc(&p, (size_t)NBIG * 2 * 8));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(p, 0, (size_t)NBIG * 2 * 8));
    printf("(1) the same S bytes again and again (plain | non-temporal loads)\n");
    for (long long mb : {8LL, 16LL, 32LL, 64LL, 96LL, 128LL, 192LL, 256LL, 384LL, 512LL, 1024LL, 2048LL}) {
        const long long n = mb << 17;
        const int reps = (int)std::max(20LL, 40960 / mb);
        const float t0 = run<0, 0>(p, n, 0, sink, reps), t1 = run<1, 1>(p, n, 0, sink, reps);
        printf("  S = %5lld MB: %8.1f | %8.1f GB/s\n", mb, n * 8 / (t0 * 1e6), n * 8 / (t1 * 1e6));
    }
    printf("(2) 3.2 GB stream + the run LAG behind (first read / re-read policy: nt/plain, plain/plain, nt/nt)\n");
    const float s1 = run<1, 1>(p, NBIG, 0, sink, 20), s2 = run<1, 1>(p, NBIG * 2, 0, sink, 10);
    printf("  plain stream of 3.2 GB: %.4f ms (%.0f GB/s); of 6.4 GB: %.4f ms (%.0f GB/s)\n", s1, NBIG * 8 / (s1 * 1e6), s2,
           NBIG * 16 / (s2 * 1e6));
    for (long long mb : {1LL, 4LL, 16LL, 32LL, 64LL, 128LL, 192LL, 256LL, 512LL}) {
        const long long lag = (mb << 17) / RUN * RUN;
        const float a = run<1, 0>(p, NBIG, lag, sink, 20), b = run<0, 0>(p, NBIG, lag, sink, 20), c = run<1, 1>(p, NBIG, lag, sink, 20);
        printf("  LAG = %4lld MB: %.4f | %.4f | %.4f ms   (requested bytes / time: %.0f | %.0f | %.0f GB/s)\n", mb, a, b, c,
               NBIG * 16 / (a * 1e6), NBIG * 16 / (b * 1e6), NBIG * 16 / (c * 1e6));
    }
    return 0;
}
