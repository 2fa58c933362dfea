// alloc_lab.cpp -- does the allocator decide how fast a 6.4 GB block streams?  (lab, not product)
// hipcc -O3 --offload-arch=gfx950 -o /tmp/alloc_lab tools/lab/alloc_lab.cpp && /tmp/alloc_lab
// Reads each block front to back the way the SpMV reads K (one wavefront per contiguous 124 KB run,
// 512 B per wave-instruction, non-temporal) and prints the rate for blocks obtained from
//   hipMalloc | hipExtMallocWithFlags(hipDeviceMallocContiguous) | hipMemCreate+hipMemMap (VMM) | hipMallocAsync
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

constexpr long long RUN = 15552;
__global__ void __launch_bounds__(256) k_probe(const double *__restrict__ p, long long n, double *sink) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long b0 = wave * RUN;
    if (b0 >= n) return;
    const long long b1 = b0 + RUN < n ? b0 + RUN : n;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    long long i = b0 + lane;
    for (; i + 192 < b1; i += 256) {
        a0 += __builtin_nontemporal_load(p + i);
        a1 += __builtin_nontemporal_load(p + i + 64);
        a2 += __builtin_nontemporal_load(p + i + 128);
        a3 += __builtin_nontemporal_load(p + i + 192);
    }
    for (; i < b1; i += 64) a0 += __builtin_nontemporal_load(p + i);
    const double s = (a0 + a1) + (a2 + a3);
    if (s == 0.1234567890123) sink[0] = s;
}

static double rate(const void *p, size_t bytes, double *sink) {
    const long long n = (long long)(bytes / 8);
    const unsigned grid = (unsigned)(((n + RUN - 1) / RUN + 3) / 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> t;
    for (int r = 0; r < 6; r++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_probe, dim3(grid), dim3(256), 0, 0, (const double *)p, n, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (r) t.push_back(ms);
    }
    hipEventDestroy(a); hipEventDestroy(b);
    std::sort(t.begin(), t.end());
    return bytes / (t[t.size() / 2] * 1e-3) / 1e9;
}

int main() {
    const size_t bytes = (size_t)6430 << 20;  // ~ K's value array at 148^3
    double *sink; hipMalloc(&sink, 64);
    std::vector<void *> keep;                 // odd-sized blocks kept alive to perturb placement
    for (int mode = 0; mode < 4; mode++) {
        const char *name[] = {"hipMalloc", "contiguous", "vmm(1 handle)", "hipMallocAsync"};
        printf("%-16s", name[mode]);
        for (int i = 0; i < 6; i++) {
            void *p = nullptr;
            hipMemGenericAllocationHandle_t h{};
            bool ok = false;
            size_t sz = bytes;
            if (mode == 0) ok = hipMalloc(&p, bytes) == hipSuccess;
            else if (mode == 1) ok = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous) == hipSuccess;
            else if (mode == 2) {
                hipMemAllocationProp prop{};
                prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
                size_t gran = 0;
                if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && gran) {
                    sz = (bytes + gran - 1) / gran * gran;
                    hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
                    ok = hipMemCreate(&h, sz, &prop, 0) == hipSuccess && hipMemAddressReserve(&p, sz, 0, nullptr, 0) == hipSuccess &&
                         hipMemMap(p, sz, 0, h, 0) == hipSuccess && hipMemSetAccess(p, sz, &acc, 1) == hipSuccess;
                }
            } else ok = hipMallocAsync(&p, bytes, 0) == hipSuccess && hipStreamSynchronize(0) == hipSuccess;
            if (!ok) { (void)hipGetLastError(); printf("  failed"); continue; }
            printf("  %6.0f", rate(p, bytes, sink));
            fflush(stdout);
            if (mode == 2) { hipMemUnmap(p, sz); hipMemAddressFree(p, sz); hipMemRelease(h); }
            else if (mode == 3) { hipFreeAsync(p, 0); hipStreamSynchronize(0); }
            else hipFree(p);
            void *q = nullptr;
            if (hipMalloc(&q, ((size_t)96 << 20) + 4096 * (size_t)(7 * i + 3)) == hipSuccess) keep.push_back(q);
        }
        printf("   GB/s\n");
    }
    return 0;
}
