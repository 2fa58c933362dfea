// ipc_probe.cpp -- lab: does a stream wait (hipStreamWaitValue64) on DEVICE memory see an arrival count that a
// kernel of ANOTHER PROCESS adds through a HIP IPC mapping, and does the data stored in front of it arrive?
// (round 3: the process-per-GPU form of the peer-to-peer exchanges.)  Both processes use GPU 0.
//   hipcc --offload-arch=gfx950 -O2 -o ipc_probe ipc_probe.cpp ; ./ipc_probe
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%d] FAIL %s -> %s\n", (int)getpid(), #x, hipGetErrorString(e_)); fflush(stdout); } } while (0)

__global__ void k_signal(double *data, double v, unsigned long long *sig) {
    if (threadIdx.x == 0) {
        __hip_atomic_store(data, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __atomic_thread_fence(__ATOMIC_RELEASE);
        __hip_atomic_fetch_add(sig, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void k_consume(const double *data, double *out) {
    if (threadIdx.x == 0) *out = __hip_atomic_load(data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_poll(const unsigned long long *flag, unsigned long long want, const double *data, double *out) {
    if (threadIdx.x == 0) {
        long n = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want && n < 400000000L) { n++; __builtin_amdgcn_s_sleep(4); }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        *out = __hip_atomic_load(data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    int a2b[2], b2a[2];
    if (pipe(a2b) || pipe(b2a)) return 1;
    const pid_t pid = fork();   // before any HIP call
    if (pid == 0) {             // ---- B: the producer
        hipIpcMemHandle_t h[2];
        for (int kind = 0; kind < 2; kind++) {
            if (read(a2b[0], &h[kind], sizeof(h[kind])) != (ssize_t)sizeof(h[kind])) return 2;
        }
        void *blk[2] = {nullptr, nullptr};
        for (int kind = 0; kind < 2; kind++) CK(hipIpcOpenMemHandle(&blk[kind], h[kind], hipIpcMemLazyEnablePeerAccess));
        hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        char ok = 1;
        if (write(b2a[1], &ok, 1) != 1) return 2;
        for (;;) {
            int cmd[2];
            if (read(a2b[0], cmd, sizeof(cmd)) != (ssize_t)sizeof(cmd) || cmd[0] < 0) break;
            char *base = (char *)blk[cmd[0]];
            hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, st, (double *)base, 100.0 + cmd[1], (unsigned long long *)(base + 128));
            CK(hipStreamSynchronize(st));
            if (write(b2a[1], &ok, 1) != 1) break;
        }
        for (int kind = 0; kind < 2; kind++) if (blk[kind]) CK(hipIpcCloseMemHandle(blk[kind]));
        return 0;
    }
    // ---- A: owner of the memory, the consumer
    void *blk[2] = {nullptr, nullptr};
    const char *name[2] = {"fine-grained device memory", "plain hipMalloc"};
    CK(hipExtMallocWithFlags(&blk[0], 4096, hipDeviceMallocFinegrained));
    CK(hipMalloc(&blk[1], 4096));
    hipIpcMemHandle_t h[2];
    for (int kind = 0; kind < 2; kind++) {
        CK(hipMemset(blk[kind], 0, 4096));
        const hipError_t e = hipIpcGetMemHandle(&h[kind], blk[kind]);
        printf("hipIpcGetMemHandle(%s): %s\n", name[kind], hipGetErrorString(e));
        if (write(a2b[1], &h[kind], sizeof(h[kind])) != (ssize_t)sizeof(h[kind])) return 1;
    }
    char ok;
    if (read(b2a[0], &ok, 1) != 1) return 1;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    double *out; CK(hipHostMalloc((void **)&out, 64, hipHostMallocDefault));
    for (int mode = 0; mode < 2; mode++)          // 0: hipStreamWaitValue64, 1: polling kernel
        for (int kind = 0; kind < 2; kind++) {
            char *base = (char *)blk[kind];
            unsigned long long have = 0;
            CK(hipMemcpy(&have, base + 128, 8, hipMemcpyDeviceToHost));
            int early = 0, wrong = 0, stuck = 0;
            double lat = 0;
            for (int rep = 0; rep < 20; rep++) {
                *out = -1;
                if (mode == 0) {
                    const hipError_t e = hipStreamWaitValue64(st, base + 128, have + 1, hipStreamWaitValueGte, ~0ULL);
                    if (e != hipSuccess) { printf("  WaitValue64 refused: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); stuck = 20; break; }
                    hipLaunchKernelGGL(k_consume, dim3(1), dim3(64), 0, st, (const double *)base, out);
                } else
                    hipLaunchKernelGGL(k_poll, dim3(1), dim3(64), 0, st, (const unsigned long long *)(base + 128), have + 1, (const double *)base, out);
                usleep(3000);
                if (hipStreamQuery(st) == hipSuccess) early++;
                int cmd[2] = {kind, rep};
                const double t0 = now();
                if (write(a2b[1], cmd, sizeof(cmd)) != (ssize_t)sizeof(cmd)) return 1;
                if (read(b2a[0], &ok, 1) != 1) return 1;          // B's kernel has completed
                const double t1 = now();
                while (hipStreamQuery(st) != hipSuccess && now() - t1 < 3.0) {}
                if (hipStreamQuery(st) != hipSuccess) {
                    stuck++;
                    unsigned long long v = have + 1;              // release the wait from the host
                    CK(hipMemcpy(base + 128, &v, 8, hipMemcpyHostToDevice));
                    CK(hipStreamSynchronize(st));
                } else {
                    lat += now() - t1;
                    if (*out != 100.0 + rep) wrong++;
                }
                (void)t0;
                CK(hipMemcpy(&have, base + 128, 8, hipMemcpyDeviceToHost));
            }
            printf("%-22s counter in %-28s: early %2d  stuck %2d  wrong data %2d  (of 20; consumer done %.0f us after the producer's kernel on average)\n",
                   mode == 0 ? "hipStreamWaitValue64" : "polling kernel", name[kind], early, stuck, wrong, lat / 20 * 1e6);
            fflush(stdout);
        }
    int cmd[2] = {-1, 0};
    if (write(a2b[1], cmd, sizeof(cmd)) != (ssize_t)sizeof(cmd)) return 1;
    int status = 0;
    waitpid(pid, &status, 0);
    printf("ipc probe done (child status %d)\n", status);
    return 0;
}
