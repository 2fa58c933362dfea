// vec_group_lab.cpp -- do the CG's vector kernels pay for writing into the memory group they read from?  (lab, not product)
// hipcc -O3 --offload-arch=gfx950 -o /tmp/vec_group_lab tools/lab/vec_group_lab.cpp && /tmp/vec_group_lab
// tools/lab/spmv_steps_lab.cpp (sweep) shows the SpMV 3 % or 15 % slower depending only on where the vector it WRITES lies
// relative to the matrix it reads.  The vector kernels read and write 79 MB vectors that the library allocates one after the
// other (one group).  Here: r' = r - a v (k_step's traffic: two reads, one write, 10 M doubles each) with the operands in
// spacer blocks of 8 GB allocated one after the other; in place, out of place into the same spacer, out of place into
// every other spacer.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NT>
__global__ void __launch_bounds__(256) k_axpy(long long n, const double *__restrict__ r, const double *__restrict__ v, double *o, double a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const double t = r[i] - a * v[i];
        if (NT) __builtin_nontemporal_store(t, o + i);
        else o[i] = t;
    }
}

template <int NT>
static float run(long long n, const double *r, const double *v, double *o, int reps) {
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_axpy<NT>), dim3(grid), dim3(256), 0, 0, n, r, v, o, 0.5);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_axpy<NT>), dim3(grid), dim3(256), 0, 0, n, r, v, o, 0.5);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms / reps * 1e3f;
}

// cold form: a 6.4 GB read sweep between the launches (what the SpMV leaves behind in the CG), events around the axpy alone
__global__ void __launch_bounds__(256) k_sweep(const double *__restrict__ p, long long n, double *sink) {
    const long long stride = (long long)gridDim.x * 256;
    double a = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) a += __builtin_nontemporal_load(p + i);
    if (a == 0.1234567890123) sink[0] = a;
}
template <int NT>
static float run_cold(long long n, const double *r, const double *v, double *o, const double *big, double *sink, int reps) {
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float sum = 0;
    for (int i = 0; i < reps + 2; i++) {
        hipLaunchKernelGGL(k_sweep, dim3(256 * 24), dim3(256), 0, 0, big, (long long)800 << 20, sink);
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL((k_axpy<NT>), dim3(grid), dim3(256), 0, 0, n, r, v, o, 0.5);
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
    const long long n = 9857244;   // the reduced system of the 148^3 cube
    std::vector<double *> sp;
    for (int i = 0; i < 28; i++) {
        double *q = nullptr;
        if (hipMalloc(&q, (size_t)8 << 30) != hipSuccess) { (void)hipGetLastError(); break; }
        CK(hipMemset(q, 0, (size_t)n * 8 * 4));
        sp.push_back(q);
    }
    const long long pad = (n + 511) & ~511LL;
    double *r = sp[0], *v = sp[0] + pad, *o = sp[0] + 2 * pad;
    printf("r' = r - a v, %lld doubles (237 MB per launch), us per launch (plain | non-temporal store)\n", n);
    printf("  in place (r, v, r' = r in spacer 0):              %.2f | %.2f\n", run<0>(n, r, v, r, 200), run<1>(n, r, v, r, 200));
    printf("  out of place, all three in spacer 0:               %.2f | %.2f\n", run<0>(n, r, v, o, 200), run<1>(n, r, v, o, 200));
    for (size_t j = 1; j < sp.size(); j++)
        printf("  r, v in spacer 0, r' in spacer %2zu:                  %.2f | %.2f     (r in 0, v in %2zu, in place: %.2f)\n", j,
               run<0>(n, r, v, sp[j], 200), run<1>(n, r, v, sp[j], 200), j, run<1>(n, r, sp[j], r, 200));
    printf("COLD (a 6.4 GB read sweep of spacer %zu before every launch), non-temporal store:\n", sp.size() - 1);
    const double *big = sp.back();
    printf("  in place, r and v in spacer 0:                     %.2f\n", run_cold<1>(n, r, v, r, big, sp[1], 20));
    printf("  out of place, all three in spacer 0:               %.2f\n", run_cold<1>(n, r, v, o, big, sp[1], 20));
    for (size_t j = 1; j + 1 < sp.size(); j += 2)
        printf("  r' in spacer %2zu: %.2f     v in spacer %2zu, in place: %.2f     r and v in spacer %2zu, in place: %.2f\n", j,
               run_cold<1>(n, r, v, sp[j] + pad, big, sp[1], 20), j, run_cold<1>(n, r, sp[j] + pad, r, big, sp[1], 20), j,
               run_cold<1>(n, sp[j] + pad, sp[j] + 2 * pad, sp[j] + pad, big, sp[1], 20));
    return 0;
}
