// waitvalue_probe.cpp -- lab: what hipStreamWaitValue64 accepts and costs on this box, and whether
// streams that share a hardware queue can deadlock on it (round 3: peer-to-peer reductions of the
// one-process multi-GPU handle).   hipcc --offload-arch=gfx950 -O2 -o waitvalue_probe waitvalue_probe.cpp
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) { printf("FAIL %s -> %s\n", #x, hipGetErrorString(e_)); } \
    } while (0)

__global__ void k_signal(double *data, double v, unsigned long long *sig) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        __hip_atomic_store(data, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __atomic_thread_fence(__ATOMIC_RELEASE);   // system scope
        __hip_atomic_fetch_add(sig, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void k_consume(const double *data, double *out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = __hip_atomic_load(data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_spin(const unsigned long long *flag, unsigned long long want, double *out, const double *data) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        long n = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want && n < 200000000L) { n++; __builtin_amdgcn_s_sleep(8); }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        *out = __hip_atomic_load(data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void k_busy(double *p, int n) {
    double a = p[threadIdx.x];
    for (int i = 0; i < n; i++) a = a * 1.0000001 + 1e-9;
    p[threadIdx.x] = a;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const int nstreams = argc > 1 ? atoi(argv[1]) : 8;
    int attr = -1;
    CK(hipDeviceGetAttribute(&attr, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("CanUseStreamWaitValue = %d; GPU_MAX_HW_QUEUES = %s\n", attr, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset)");

    // --- what memory does the wait accept?
    unsigned long long *sig = nullptr, *fine = nullptr, *plain = nullptr, *hostp = nullptr;
    hipError_t e;
    e = hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory);
    printf("alloc signal memory: %s ptr %p\n", hipGetErrorString(e), (void *)sig);
    e = hipExtMallocWithFlags((void **)&fine, 4096, hipDeviceMallocFinegrained);
    printf("alloc fine-grained device memory: %s\n", hipGetErrorString(e));
    CK(hipMalloc((void **)&plain, 4096));
    CK(hipHostMalloc((void **)&hostp, 4096, hipHostMallocDefault));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CK(hipMemset(fine, 0, 4096)); CK(hipMemset(plain, 0, 4096));
    hostp[0] = 0;
    if (sig) *sig = 0;   // host store to the signal's value
    CK(hipDeviceSynchronize());
    struct { const char *name; unsigned long long *p; } cand[4] = {{"signal", sig}, {"fine-grained", fine}, {"plain hipMalloc", plain}, {"hipHostMalloc", hostp}};
    double *data, *out;
    CK(hipExtMallocWithFlags((void **)&data, 4096, hipDeviceMallocFinegrained));
    CK(hipHostMalloc((void **)&out, 64, hipHostMallocDefault));
    double *busy; CK(hipMalloc((void **)&busy, 4096)); CK(hipMemset(busy, 0, 4096));
    for (auto &c : cand) {
        if (!c.p) continue;
        // already satisfied wait (value 0 >= 0): does the API take this pointer?
        e = hipStreamWaitValue64(sb, c.p, 0, hipStreamWaitValueGte, ~0ULL);
        printf("WaitValue64 on %-16s: %s", c.name, hipGetErrorString(e));
        if (e != hipSuccess) { (void)hipGetLastError(); printf("\n"); continue; }
        e = hipStreamSynchronize(sb);
        printf("; sync %s\n", hipGetErrorString(e));
        // real hand-off: B waits for >= 1, A's kernel stores data then adds 1
        *out = -1;
        unsigned long long base = 0;
        for (int rep = 0; rep < 3; rep++) {
            const double t0 = now();
            CK(hipStreamWaitValue64(sb, c.p, base + 1, hipStreamWaitValueGte, ~0ULL));
            hipLaunchKernelGGL(k_consume, dim3(1), dim3(64), 0, sb, data, out);
            usleep(2000);   // B is waiting now
            const bool early = hipStreamQuery(sb) == hipSuccess;
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, sa, data, 42.0 + rep, c.p);
            double tq = now();
            while (hipStreamQuery(sb) != hipSuccess && now() - tq < 5.0) {}
            const bool ok = hipStreamQuery(sb) == hipSuccess;
            if (!ok) { printf("   rep %d: consumer still waiting after 5 s (releasing by host store)\n", rep);
                       if (c.p == hostp || c.p == sig) *c.p = base + 1000; else { unsigned long long v = base + 1000; CK(hipMemcpy(c.p, &v, 8, hipMemcpyHostToDevice)); }
                       CK(hipStreamSynchronize(sb)); base += 1000; continue; }
            CK(hipStreamSynchronize(sa));
            printf("   rep %d: consumer ran early=%d, saw data %.1f (want %.1f), total %.3f ms\n", rep, (int)early, *out, 42.0 + rep, (now() - t0) * 1e3);
            base += 1;
        }
        // latency: A: [busy kernel, signal]; B: [wait, consume]; event after A's signal kernel vs event after B's consume
        hipEvent_t ea, eb0, eb1;
        CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb0)); CK(hipEventCreate(&eb1));
        float acc = 0; int cnt = 0;
        for (int rep = 0; rep < 20; rep++) {
            CK(hipStreamWaitValue64(sb, c.p, base + 1, hipStreamWaitValueGte, ~0ULL));
            hipLaunchKernelGGL(k_consume, dim3(1), dim3(64), 0, sb, data, out);
            CK(hipEventRecord(eb1, sb));
            hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, sa, busy, 20000);
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, sa, data, 1.0, c.p);
            CK(hipEventRecord(ea, sa));
            double tq = now();
            while (hipEventQuery(eb1) != hipSuccess && now() - tq < 5.0) {}
            if (hipEventQuery(eb1) != hipSuccess) { printf("   latency rep stuck\n"); break; }
            CK(hipEventSynchronize(ea));
            float ms = 0;
            if (hipEventElapsedTime(&ms, ea, eb1) == hipSuccess) { acc += ms; cnt++; }
            base += 1;
        }
        if (cnt) printf("   signal-kernel end -> consumer end: %.1f us (avg of %d)\n", acc / cnt * 1e3, cnt);
    }
    // --- a spin kernel instead of the stream wait (flag in fine-grained device memory)
    {
        CK(hipMemset(fine, 0, 4096));
        hipEvent_t ea, eb1; CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb1));
        float acc = 0; int cnt = 0; unsigned long long base = 0;
        for (int rep = 0; rep < 20; rep++) {
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, sb, fine, base + 1, out, data);
            CK(hipEventRecord(eb1, sb));
            hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, sa, busy, 20000);
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, sa, data, 7.0, fine);
            CK(hipEventRecord(ea, sa));
            CK(hipEventSynchronize(eb1)); CK(hipEventSynchronize(ea));
            float ms = 0;
            if (hipEventElapsedTime(&ms, ea, eb1) == hipSuccess) { acc += ms; cnt++; }
            base += 1;
        }
        printf("spin kernel on a fine-grained flag: signal-kernel end -> consumer end %.1f us (avg of %d), data %.1f\n", acc / cnt * 1e3, cnt, *out);
    }
    // --- do streams share hardware queues so that a blocked wait blocks a producer?
    // stream i: [wait sig[i] >= 1, kernel adds 1 to sig[i+1]]; enqueued last stream first; then sig[0] is raised.
    {
        const int n = nstreams;
        std::vector<hipStream_t> st(n);
        std::vector<unsigned long long *> sg(n + 1);
        for (int i = 0; i < n; i++) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
        bool ok = true;
        for (int i = 0; i <= n; i++) { if (hipExtMallocWithFlags((void **)&sg[i], 8, hipMallocSignalMemory) != hipSuccess) { ok = false; break; } *sg[i] = 0; }
        if (ok) {
            for (int i = n - 1; i >= 0; i--) {
                CK(hipStreamWaitValue64(st[i], sg[i], 1, hipStreamWaitValueGte, ~0ULL));
                hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, st[i], data, (double)i, sg[i + 1]);
            }
            const double t0 = now();
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, sa, data, -1.0, sg[0]);
            bool done = false;
            while (now() - t0 < 5.0) { if (*(volatile unsigned long long *)sg[n] >= 1) { done = true; break; } }
            printf("chain of %d streams (reverse enqueue order): %s after %.3f ms\n", n, done ? "COMPLETED" : "DEADLOCK (streams share a hardware queue)", (now() - t0) * 1e3);
            if (!done) for (int i = 0; i <= n; i++) *sg[i] = 1000;   // release every wait from the host
            for (int i = 0; i < n; i++) CK(hipStreamSynchronize(st[i]));
            printf("chain drained\n");
        } else printf("chain test skipped: no signal memory\n");
    }
    CK(hipDeviceSynchronize());
    printf("probe done\n");
    return 0;
}
