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
    return 0;
}
