// stream_lab.cpp -- lab: what read rate does the memory system give a kernel that streams a large block the way the
// SpMV streams the matrix (one wavefront per contiguous chunk, a few "rows" of one wave-wide load each per trip),
// as a function of the bytes per lane (8 / 16), of the number of lanes that take part in a row (64 / 48 / 32: the
// ragged streams of round 3), of how the idle lanes are silenced (exec mask / re-reading the last active lane's
// address) and of where the next row starts (padded to the full wave width / back to back)?
//   hipcc -O3 --offload-arch=gfx950 -o stream_lab stream_lab.cpp ; ./stream_lab [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double dbl2 __attribute__((ext_vector_type(2)));   // 16 B per lane: one global_load_dwordx4
template <typename T> __device__ __forceinline__ double fold(T v);
template <> __device__ __forceinline__ double fold<double>(double v) { return v; }
template <> __device__ __forceinline__ double fold<dbl2>(dbl2 v) { return v.x + v.y; }

// T: double (8 B per lane) or dbl2 (16 B); ACT: lanes that take part; MODE 0: idle lanes masked, 1: idle lanes
// re-read the last active lane; DENSE: rows back to back (ACT * sizeof(T) apart) instead of 64 * sizeof(T); ROWS rows per trip
template <typename T, int ACT, int MODE, bool DENSE, int ROWS, bool NT>
__global__ void __launch_bounds__(256) k_stream(const T *base, long long rows_per_wave, double *out) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    constexpr long long STRIDE = DENSE ? ACT : 64;
    const T *p = base + wave * rows_per_wave * STRIDE;
    const bool act = lane < ACT;
    const int l = act ? lane : ACT - 1;
    double s = 0;
    for (long long r = 0; r + ROWS <= rows_per_wave; r += ROWS) {
        T v[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; j++) {
            const T *q = p + (r + j) * STRIDE;
            if (MODE == 0) {
                if (act) v[j] = NT ? __builtin_nontemporal_load(q + lane) : q[lane];
                else v[j] = T{};
            } else
                v[j] = NT ? __builtin_nontemporal_load(q + l) : q[l];
        }
#pragma unroll
        for (int j = 0; j < ROWS; j++) s += fold<T>(v[j]);
    }
    if (s == 12345.678) out[wave] = s;   // never true: keeps the loads alive
}

template <typename T, int ACT, int MODE, bool DENSE, int ROWS, bool NT>
double run(const void *buf, size_t bytes, double *out, const char *name) {
    const long long chunk = 128 << 10;                               // bytes a wave covers (of padded or dense rows)
    const long long row_bytes = (DENSE ? ACT : 64) * (long long)sizeof(T);
    const long long rows_per_wave = chunk / row_bytes / ROWS * ROWS;
    const long long waves = (long long)(bytes / (rows_per_wave * row_bytes)) / 4 * 4;
    const double useful = (double)waves * rows_per_wave * ACT * sizeof(T);
    const double touched = (double)waves * rows_per_wave * row_bytes;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> t;
    for (int rep = 0; rep < 7; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_stream<T, ACT, MODE, DENSE, ROWS, NT>), dim3((unsigned)(waves / 4)), dim3(256), 0, 0, (const T *)buf, rows_per_wave, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep >= 2) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const double ms = t[t.size() / 2];
    printf("%-64s %8.3f ms  useful %7.0f GB/s  span %7.0f GB/s  (%.2f GB useful, %lld rows of %lld B per wave)\n", name, ms, useful / ms / 1e6,
           touched / ms / 1e6, useful / 1e9, rows_per_wave, (long long)(ACT * sizeof(T)));
    fflush(stdout);
    return ms;
}

int main(int argc, char **argv) {
    const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 6.0) * (1ull << 30);
    void *buf; double *out;
    CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    CK(hipMalloc((void **)&out, 64 << 20));
    printf("streaming read of a %.1f GiB block, one wavefront per 128-KB chunk, 9 rows per trip unless noted\n", bytes / 1073741824.0);
#define RUN(T, ACT, MODE, DENSE, ROWS, NT) run<T, ACT, MODE, DENSE, ROWS, NT>(buf, bytes, out, #T " lanes " #ACT " mode " #MODE " dense " #DENSE " rows/trip " #ROWS " nt " #NT)
    RUN(double, 64, 1, false, 9, true);
    RUN(double, 64, 1, false, 9, false);
    RUN(double, 64, 1, false, 18, true);
    RUN(dbl2, 64, 1, false, 5, true);
    RUN(dbl2, 64, 1, false, 9, true);
    RUN(dbl2, 64, 1, false, 9, false);
    RUN(double, 48, 0, false, 9, true);    // masked, rows padded to 512 B: the "masked" build
    RUN(double, 48, 1, true, 9, true);     // ragged streams: 48 lanes, rows back to back, idle lanes re-read
    RUN(double, 48, 0, true, 9, true);     // same, idle lanes masked
    RUN(double, 48, 1, true, 9, false);
    RUN(double, 32, 0, false, 9, true);
    RUN(double, 32, 1, true, 9, true);
    RUN(double, 32, 0, true, 9, true);
    RUN(dbl2, 32, 0, true, 9, true);    // 32 lanes x 16 B = the same 512-B rows from half a wave
    RUN(double, 64, 1, false, 9, true);    // the first one again (drift)
    return 0;
}
