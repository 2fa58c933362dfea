// spmv_steps_lab.cpp -- where does the SpMV lose its 6-7 % against a pure stream?  (lab, not product)
// hipcc -O3 --offload-arch=gfx950 -o /tmp/spmv_steps_lab tools/lab/spmv_steps_lab.cpp && /tmp/spmv_steps_lab [n]
// A synthetic BSELL-64 operator of the n^3 cube's shape (27 slots per slice of 64 block rows, 9 fp64 values per block in
// [slot][9][64] order, one 16-bit column offset per block two to a dword, columns = row + the 27 offsets of the node
// stencil, clamped) is read by kernels that add the SpMV's ingredients one at a time:
//   S0  the value stream alone, one wavefront per slice (what tools/lab/mall_lab.cpp calls the plain stream)
//   S1  + the packed column words (a second, 36 x thinner stream from another array)
//   S2  + the gather of x (3 doubles per block at 24 B stride across lanes, from a 3 n^3-double vector) and the FMAs
//   S3  + y stored (non-temporal) -- the SpMV's traffic, identity workgroup mapping
//   S4  = S3 with the XCD-chunked workgroup mapping of k_spmv
//   S5  = S4 with the column word of the NEXT trip loaded one trip ahead
//   S6  = S4 with y turned through LDS: three stores of 64 CONSECUTIVE doubles per wave instead of three of 8 B at 24 B stride
//   S7  = S6 with plain stores;  S8 = S4 with plain stores
//   S9  = S4 with every wave taking 4 consecutive slices one after the other (a quarter of the workgroups): the stores of
//         one slice are in flight while the next is read -- does a wave that ends on a store hold its slot for long?
//   S10 = S4 with y written into a 1.5 MB window (slice mod 1024): the stores are issued, HBM sees next to none of them
//   S11 = S9 with 8 slices per wave
//   S12 = S4 with PLAIN stores into the 1.5 MB window (the L2 absorbs them: what the store instructions themselves cost)
//   S13 = S4 storing y0 only (a third of the bytes, the same lines);  S14 = S4 with only the even slices storing
//   S15 = S4 with three non-temporal LOADS of y in place of the stores (a 79 MB read stream instead of a write stream)
//   S16 / S17 = S4 with y in fine-grained / uncached memory (hipExtMallocWithFlags)
//   S18 = S6 (full lines through LDS) with the 1.5 KB pieces of y TRANSPOSED: slice s writes piece (s mod 128) * (nslices / 128) + s / 128
//   S21-S24 = S6 with the cache-policy bits of the store instruction set by hand: sc0 | sc1 | sc0 sc1 | sc0 sc1 nt
//   S25 = S6 with the piece of slice s stored INSIDE the value array, right behind the slice's own values (slices 1.5 KB longer)
//   S26 = S2 + the same three stores issued at the START of the wave (zeros): same write traffic, but no wave ENDS on a store
//   S27 = S2 + the stores after the 7th of the 13 trips (the running sums)
//   S31 = S4 + every wave times (wall_clock64, 10 ns ticks) from its stores to their acknowledgement (s_waitcnt vmcnt(0)) and from
//         its first load to the end; S32 = the same with three LOADS of y in place of the stores (S15)
//   S33 = S4 + the product's p.Ap epilogue: x of the own rows read again, a workgroup sum through LDS behind __syncthreads, one
//         partial stored per workgroup;  S34 = the same sum without the barrier (the last of the four waves to arrive adds up);
//   S35 = S4 with the slot bases read by per-lane loads through a lane-selected pointer (what the two-base loop compiles to)
//   S36 = S4 with the gather of x replaced by three lane-contiguous loads of the same 1.5 KB window (wrong products: what does the
//         FORM of the gather cost?)
//   S19 = S6 with the pieces at a multiplicative hash of s;  S20 = S6 with piece s ^ 1 ... (neighbours swapped: control)
// The numbers are not products (values are zeros + noise): only times matter.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int W = 27;               // slots per slice
constexpr int STRIDE = 9 * 64;      // doubles per slot

template <int POL>
__device__ __forceinline__ void st_policy(double *p, double v) {
    if (POL == 21) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else if (POL == 22) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else if (POL == 23) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else if (POL == 24) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    else __builtin_nontemporal_store(v, p);
}

template <int STEP>
__global__ void __launch_bounds__(256) k_steps(int nslices, long long nb, const double *__restrict__ vals,
                                               const uint32_t *__restrict__ cw, const int *__restrict__ base,
                                               const double *__restrict__ x, double *__restrict__ y, double *sink) {
    __shared__ double ysh[4][192];
    __shared__ double dsh[4];
    __shared__ int arrived;
    if (STEP == 34) { if (threadIdx.x == 0) arrived = 0; __syncthreads(); }
    const int lane = threadIdx.x & 63;
    long long bid = blockIdx.x;
    if (STEP >= 4) {
        const long long win = 256, grp = bid / win, within = bid - grp * win;
        if ((grp + 1) * win <= (long long)gridDim.x) bid = grp * win + (within & 7) * 32 + (within >> 3);
    }
    constexpr int SPW = STEP == 9 ? 4 : STEP == 11 ? 8 : 1;
    for (int rep = 0; rep < SPW; rep++) {
    const long long slice = (bid * 4 + (threadIdx.x >> 6)) * SPW + rep;
    if (slice >= nslices) return;
    const long long t_begin = (STEP == 31 || STEP == 32) ? wall_clock64() : 0;
    const long long SL = W * STRIDE + (STEP == 25 ? 192 : 0);
    const double *vp = vals + slice * SL + lane;
    const uint32_t *cq = cw + slice * (long long)(14 * 64) + lane;       // 14 pair words per slice (27 slots)
    const int *bp = base + slice * 28;
    double y0 = 0, y1 = 0, y2 = 0;
    if (STEP == 26) {
        const long long row0 = slice * 64 + lane;
        if (row0 < nb) {
            __builtin_nontemporal_store(0.0, y + 3 * row0);
            __builtin_nontemporal_store(0.0, y + 3 * row0 + 1);
            __builtin_nontemporal_store(0.0, y + 3 * row0 + 2);
        }
    }
    uint32_t wnext = 0;
    if (STEP >= 5) wnext = __builtin_nontemporal_load(cq);
    for (int k = 0; k + 1 < W; k += 2) {
        double a[9], b[9];
        uint32_t wd = 0;
        if (STEP >= 5) { wd = wnext; wnext = __builtin_nontemporal_load(cq + 64); }   // (the 14th word exists: 27 = 13 pairs + 1)
        else if (STEP >= 1) wd = __builtin_nontemporal_load(cq);
#pragma unroll
        for (int j = 0; j < 9; j++) a[j] = __builtin_nontemporal_load(vp + j * 64);
#pragma unroll
        for (int j = 0; j < 9; j++) b[j] = __builtin_nontemporal_load(vp + STRIDE + j * 64);
        if (STEP >= 2) {
            const int *bl = (STEP == 35 && ((0x5555555555555555ull >> lane) & 1ull) && lane > 64) ? bp + 1 : bp;   // never taken, not provable
            const long long c = (long long)(STEP == 35 ? __builtin_nontemporal_load(bl + k) : bp[k]) + (wd & 0xffffu),
                            c2 = (long long)(STEP == 35 ? __builtin_nontemporal_load(bl + k + 1) : bp[k + 1]) + (wd >> 16);
            double x0, x1, x2, z0, z1, z2;
            if (STEP == 36) {
                const long long cb0 = 3 * (c - lane > 0 ? c - lane : 0) + lane, cb2 = 3 * (c2 - lane > 0 ? c2 - lane : 0) + lane;
                x0 = x[cb0]; x1 = x[cb0 + 64]; x2 = x[cb0 + 128];
                z0 = x[cb2]; z1 = x[cb2 + 64]; z2 = x[cb2 + 128];
            } else {
                x0 = x[3 * c]; x1 = x[3 * c + 1]; x2 = x[3 * c + 2];
                z0 = x[3 * c2]; z1 = x[3 * c2 + 1]; z2 = x[3 * c2 + 2];
            }
            y0 += a[0] * x0 + a[1] * x1 + a[2] * x2;
            y1 += a[3] * x0 + a[4] * x1 + a[5] * x2;
            y2 += a[6] * x0 + a[7] * x1 + a[8] * x2;
            y0 += b[0] * z0 + b[1] * z1 + b[2] * z2;
            y1 += b[3] * z0 + b[4] * z1 + b[5] * z2;
            y2 += b[6] * z0 + b[7] * z1 + b[8] * z2;
        } else {
            const double w = STEP >= 1 ? (double)(wd & 1u) : 1.0;
            y0 += (a[0] + a[1] + a[2] + b[0] + b[1] + b[2]) * w;
            y1 += a[3] + a[4] + a[5] + b[3] + b[4] + b[5];
            y2 += a[6] + a[7] + a[8] + b[6] + b[7] + b[8];
        }
        cq += 64;
        vp += 2 * STRIDE;
        if (STEP == 27 && k == 12) {
            const long long row0 = slice * 64 + lane;
            if (row0 < nb) {
                __builtin_nontemporal_store(y0, y + 3 * row0);
                __builtin_nontemporal_store(y1, y + 3 * row0 + 1);
                __builtin_nontemporal_store(y2, y + 3 * row0 + 2);
            }
        }
    }
    {   // the 27th slot
        double a[9];
        const uint32_t wd = STEP >= 5 ? wnext : STEP >= 1 ? __builtin_nontemporal_load(cq) : 0u;
#pragma unroll
        for (int j = 0; j < 9; j++) a[j] = __builtin_nontemporal_load(vp + j * 64);
        if (STEP >= 2) {
            const long long c = (long long)bp[W - 1] + (wd & 0xffffu);
            const double x0 = x[3 * c], x1 = x[3 * c + 1], x2 = x[3 * c + 2];
            y0 += a[0] * x0 + a[1] * x1 + a[2] * x2;
            y1 += a[3] * x0 + a[4] * x1 + a[5] * x2;
            y2 += a[6] * x0 + a[7] * x1 + a[8] * x2;
        } else {
            y0 += a[0] + a[1] + a[2] + (double)(wd & 1u);
            y1 += a[3] + a[4] + a[5];
            y2 += a[6] + a[7] + a[8];
        }
    }
    const long long row = slice * 64 + lane;
    if (STEP == 6 || STEP == 7 || (STEP >= 18 && STEP <= 25)) {
        double *w = ysh[threadIdx.x >> 6];
        w[3 * lane] = y0; w[3 * lane + 1] = y1; w[3 * lane + 2] = y2;      // one wave: no barrier needed, LDS ops are in order
        long long piece = slice;
        if (STEP == 18) { const long long q = nslices / 128; if (slice < q * 128) piece = (slice & 127) * q + (slice >> 7); }
        if (STEP == 19) { const long long q = nslices & ~1023; if (slice < q) piece = (slice & ~1023LL) | ((slice * 421) & 1023); }
        if (STEP == 20) piece = slice ^ 1;
        double *yo = STEP == 25 ? const_cast<double *>(vals) + slice * SL + W * STRIDE : y + 3 * piece * 64;
        const long long lim = STEP >= 18 ? 192 : 3 * (nb - slice * 64);
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (64 * j + lane < lim) {
                if (STEP >= 21 && STEP <= 24) st_policy<STEP>(yo + 64 * j + lane, w[64 * j + lane]);
                else if (STEP != 7) __builtin_nontemporal_store(w[64 * j + lane], yo + 64 * j + lane);
                else yo[64 * j + lane] = w[64 * j + lane];
            }
    } else if (STEP == 33 || STEP == 34) {
        double d = 0;
        if (row < nb) {
            __builtin_nontemporal_store(y0, y + 3 * row);
            __builtin_nontemporal_store(y1, y + 3 * row + 1);
            __builtin_nontemporal_store(y2, y + 3 * row + 2);
            d = y0 * x[3 * row] + y1 * x[3 * row + 1] + y2 * x[3 * row + 2];
        }
        for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o);
        const int wv = threadIdx.x >> 6;
        if (STEP == 33) {
            if (lane == 0) dsh[wv] = d;
            __syncthreads();
            if (threadIdx.x == 0) sink[1024 + blockIdx.x] = (dsh[0] + dsh[1]) + (dsh[2] + dsh[3]);
        } else if (lane == 0) {
            dsh[wv] = d;
            __threadfence_block();
            if (atomicAdd(&arrived, 1) == 3) {
                __threadfence_block();
                sink[1024 + blockIdx.x] = (((volatile double *)dsh)[0] + ((volatile double *)dsh)[1]) + (((volatile double *)dsh)[2] + ((volatile double *)dsh)[3]);
            }
        }
    } else if (STEP == 31 || STEP == 32) {
        const long long t1 = wall_clock64();
        double q = 0;
        if (row < nb) {
            if (STEP == 31) {
                __builtin_nontemporal_store(y0, y + 3 * row);
                __builtin_nontemporal_store(y1, y + 3 * row + 1);
                __builtin_nontemporal_store(y2, y + 3 * row + 2);
            } else
                q = __builtin_nontemporal_load(y + 3 * row) + __builtin_nontemporal_load(y + 3 * row + 1) + __builtin_nontemporal_load(y + 3 * row + 2);
        }
        __builtin_amdgcn_s_waitcnt(0);     // vmcnt(0) expcnt(0) lgkmcnt(0)
        if (STEP == 32) asm volatile("" ::"v"(q));
        const long long t2 = wall_clock64();
        if (lane == 0) {
            long long *tm = (long long *)sink;
            tm[8 + 2 * slice] = t2 - t1;
            tm[8 + 2 * slice + 1] = t2 - t_begin;
        }
    } else if (STEP == 26 || STEP == 27) {
        if (y0 + y1 + y2 == 0.1234567890123) sink[0] = y0;
    } else if (STEP == 8) {
        if (row < nb) { y[3 * row] = y0; y[3 * row + 1] = y1; y[3 * row + 2] = y2; }
    } else if (STEP == 12) {
        const long long r2 = (slice & 1023) * 64 + lane;
        y[3 * r2] = y0; y[3 * r2 + 1] = y1; y[3 * r2 + 2] = y2;
    } else if (STEP == 13) {
        if (row < nb) __builtin_nontemporal_store(y0, y + 3 * row);
    } else if (STEP == 14) {
        if (row < nb && !(slice & 1)) {
            __builtin_nontemporal_store(y0, y + 3 * row);
            __builtin_nontemporal_store(y1, y + 3 * row + 1);
            __builtin_nontemporal_store(y2, y + 3 * row + 2);
        }
    } else if (STEP == 15) {
        if (row < nb) {
            const double q = y0 + __builtin_nontemporal_load(y + 3 * row) + __builtin_nontemporal_load(y + 3 * row + 1) +
                             __builtin_nontemporal_load(y + 3 * row + 2);
            if (q + y1 + y2 == 0.1234567890123) sink[0] = q;
        }
    } else if (STEP == 10) {
        const long long r2 = (slice & 1023) * 64 + lane;
        __builtin_nontemporal_store(y0, y + 3 * r2);
        __builtin_nontemporal_store(y1, y + 3 * r2 + 1);
        __builtin_nontemporal_store(y2, y + 3 * r2 + 2);
    } else if (STEP >= 3) {
        if (row < nb) {
            __builtin_nontemporal_store(y0, y + 3 * row);
            __builtin_nontemporal_store(y1, y + 3 * row + 1);
            __builtin_nontemporal_store(y2, y + 3 * row + 2);
        }
    } else if (y0 + y1 + y2 == 0.1234567890123) sink[0] = y0;
    }
}

// S28: PERSISTENT waves.  The grid is what the device holds at once; every wave draws slices from its XCD's counter (XCD x
// owns the 128-slice chunks x, x + 8, x + 16 ...: the chunked mapping of S4 as a work queue), the ticket of the next slice
// is drawn while the current one is read, an XCD that has run dry steals from its neighbours.  A wave's stores are
// followed by the loads of its next slice: no wave slot waits on a store, no workgroup launch between slices.
__device__ __forceinline__ long long ticket_to_slice(long long t, int xcd, long long nchunks) {
    const long long c = (t >> 7) * 8 + xcd;             // chunk
    return c < nchunks ? c * 128 + (t & 127) : -1;
}
template <int MODE>
__global__ void __launch_bounds__(256) k_persist(int nslices, long long nb, const double *__restrict__ vals,
                                                 const uint32_t *__restrict__ cw, const int *__restrict__ base,
                                                 const double *__restrict__ x, double *__restrict__ y, unsigned long long *cnt) {
    const int lane = threadIdx.x & 63;
    const long long nchunks = ((long long)nslices + 127) >> 7;
    int xcd = MODE == 1 ? (int)(blockIdx.x & 7) : (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7);
    int tries = 0;
    long long t = 0;
    if (lane == 0) t = (long long)atomicAdd(cnt + 16 * xcd, 1ull);
    t = __shfl(t, 0);
    for (;;) {
        long long slice = ticket_to_slice(t, xcd, nchunks);
        while (slice < 0 || slice >= nslices) {          // this XCD's queue is dry (or the ragged last chunk): try the next one
            if (slice < 0) { if (++tries == 8) return; xcd = (xcd + 1) & 7; }
            if (lane == 0) t = (long long)atomicAdd(cnt + 16 * xcd, 1ull);
            t = __shfl(t, 0);
            slice = ticket_to_slice(t, xcd, nchunks);
        }
        long long tn = 0;
        if (lane == 0) tn = (long long)atomicAdd(cnt + 16 * xcd, 1ull);   // the next ticket: in flight while this slice is read
        const double *vp = vals + slice * (long long)(W * STRIDE) + lane;
        const uint32_t *cq = cw + slice * (long long)(14 * 64) + lane;
        const int *bp = base + slice * 28;
        double y0 = 0, y1 = 0, y2 = 0;
        for (int k = 0; k + 1 < W; k += 2) {
            double a[9], b[9];
            const uint32_t wd = __builtin_nontemporal_load(cq);
#pragma unroll
            for (int j = 0; j < 9; j++) a[j] = __builtin_nontemporal_load(vp + j * 64);
#pragma unroll
            for (int j = 0; j < 9; j++) b[j] = __builtin_nontemporal_load(vp + STRIDE + j * 64);
            const long long c = (long long)bp[k] + (wd & 0xffffu), c2 = (long long)bp[k + 1] + (wd >> 16);
            const double x0 = x[3 * c], x1 = x[3 * c + 1], x2 = x[3 * c + 2];
            const double z0 = x[3 * c2], z1 = x[3 * c2 + 1], z2 = x[3 * c2 + 2];
            y0 += a[0] * x0 + a[1] * x1 + a[2] * x2;
            y1 += a[3] * x0 + a[4] * x1 + a[5] * x2;
            y2 += a[6] * x0 + a[7] * x1 + a[8] * x2;
            y0 += b[0] * z0 + b[1] * z1 + b[2] * z2;
            y1 += b[3] * z0 + b[4] * z1 + b[5] * z2;
            y2 += b[6] * z0 + b[7] * z1 + b[8] * z2;
            cq += 64;
            vp += 2 * STRIDE;
        }
        {
            double a[9];
            const uint32_t wd = __builtin_nontemporal_load(cq);
#pragma unroll
            for (int j = 0; j < 9; j++) a[j] = __builtin_nontemporal_load(vp + j * 64);
            const long long c = (long long)bp[W - 1] + (wd & 0xffffu);
            const double x0 = x[3 * c], x1 = x[3 * c + 1], x2 = x[3 * c + 2];
            y0 += a[0] * x0 + a[1] * x1 + a[2] * x2;
            y1 += a[3] * x0 + a[4] * x1 + a[5] * x2;
            y2 += a[6] * x0 + a[7] * x1 + a[8] * x2;
        }
        const long long row = slice * 64 + lane;
        if (row < nb) {
            __builtin_nontemporal_store(y0, y + 3 * row);
            __builtin_nontemporal_store(y1, y + 3 * row + 1);
            __builtin_nontemporal_store(y2, y + 3 * row + 2);
        }
        t = __shfl(tn, 0);
    }
}

template <int MODE>
static float run_persist(int nslices, long long nb, const double *vals, const uint32_t *cw, const int *base, const double *x, double *y,
                         unsigned long long *cnt, int reps, int wg_per_cu) {
    int per_cu = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_persist<MODE>, 256, 0);
    if (wg_per_cu > 0 && wg_per_cu < per_cu) per_cu = wg_per_cu;
    const int grid = 256 * per_cu;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; i++) {
        (void)hipMemsetAsync(cnt, 0, 8 * 16 * 8, 0);
        hipLaunchKernelGGL((k_persist<MODE>), dim3(grid), dim3(256), 0, 0, nslices, nb, vals, cw, base, x, y, cnt);
    }
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) {
        (void)hipMemsetAsync(cnt, 0, 8 * 16 * 8, 0);
        hipLaunchKernelGGL((k_persist<MODE>), dim3(grid), dim3(256), 0, 0, nslices, nb, vals, cw, base, x, y, cnt);
    }
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    printf("    (persistent grid: %d workgroups = %d per CU; the time includes a memset of the counters per launch)\n", grid, per_cu);
    return ms / reps;
}

template <int STEP>
static float run(int nslices, long long nb, const double *vals, const uint32_t *cw, const int *base, const double *x, double *y,
                 double *sink, int reps) {
    constexpr int SPW = STEP == 9 ? 4 : STEP == 11 ? 8 : 1;
    const int grid = (nslices + 4 * SPW - 1) / (4 * SPW);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_steps<STEP>), dim3(grid), dim3(256), 0, 0, nslices, nb, vals, cw, base, x, y, sink);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_steps<STEP>), dim3(grid), dim3(256), 0, 0, nslices, nb, vals, cw, base, x, y, sink);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms / reps;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 148;
    const long long e = n + 1, nb = e * e * e;
    const int nslices = (int)((nb + 63) / 64);
    const size_t nv = (size_t)nslices * (W * STRIDE + 192);
    double *vals, *x, *y, *sink;
    uint32_t *cw;
    int *base;
    // the vectors first, the matrix last: the order the library allocates in
    CK(hipMalloc(&x, (size_t)(nslices * 64LL + 64) * 3 * 8));
    CK(hipMalloc(&y, (size_t)(nslices * 64LL + 64) * 3 * 8));
    CK(hipMalloc(&sink, 64 + (size_t)nslices * 16 + 1024 * 8));
    unsigned long long *cnt;
    CK(hipMalloc(&cnt, 8 * 16 * 8));
    double *yf = nullptr, *yu = nullptr;
    if (hipExtMallocWithFlags((void **)&yf, (size_t)(nslices * 64LL + 64) * 3 * 8, hipDeviceMallocFinegrained) != hipSuccess) yf = nullptr;
    if (hipExtMallocWithFlags((void **)&yu, (size_t)(nslices * 64LL + 64) * 3 * 8, hipDeviceMallocUncached) != hipSuccess) yu = nullptr;
    (void)hipGetLastError();
    CK(hipMalloc(&cw, (size_t)nslices * 14 * 64 * 4 + 1024));
    CK(hipMalloc(&base, (size_t)nslices * 28 * 4));
    CK(hipMalloc(&vals, nv * 8));
    CK(hipMemset(vals, 0, nv * 8));
    CK(hipMemset(x, 0, (size_t)(nslices * 64LL + 64) * 3 * 8));
    // columns: row + stencil offset, clamped; per slot the smallest is the base, the rest 16-bit offsets
    std::vector<int> hb((size_t)nslices * 28, 0);
    std::vector<uint32_t> hw((size_t)nslices * 14 * 64, 0);
    long long off[W];
    {
        int k = 0;
        for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) off[k++] = (dz * e + dy) * e + dx;
    }
    for (int s = 0; s < nslices; s++)
        for (int k = 0; k < W; k++) {
            long long lo = 1LL << 60, c[64];
            for (int l = 0; l < 64; l++) {
                long long r = (long long)s * 64 + l + off[k];
                r = r < 0 ? 0 : r >= nb ? nb - 1 : r;
                c[l] = r;
                lo = std::min(lo, r);
            }
            hb[(size_t)s * 28 + k] = (int)lo;
            for (int l = 0; l < 64; l++) {
                uint32_t &w = hw[((size_t)s * 14 + k / 2) * 64 + l];
                const uint32_t o = (uint32_t)(c[l] - lo);
                w |= (k & 1) ? o << 16 : o;
            }
        }
    CK(hipMemcpy(base, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(cw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    const double gb_vals = (double)nslices * W * STRIDE * 8 / 1e9, gb_all = gb_vals + nslices * 14 * 256 / 1e9 + 2 * nb * 24 / 1e9;
    printf("cube %d^3: %lld block rows, %d slices, value stream %.3f GB, with columns, x and y %.3f GB\n", n, nb, nslices, gb_vals, gb_all);
    const char *names[37] = {"S0 values only", "S1 + column words", "S2 + x gather + FMAs", "S3 + y stored", "S4 + XCD-chunked mapping",
                            "S5 + column word one trip ahead", "S6 = S4, y through LDS, nt stores", "S7 = S6, plain stores",
                            "S8 = S4, plain strided stores", "S9 = S4, 4 slices per wave", "S10 = S4, y into a 1.5 MB window",
                            "S11 = S4, 8 slices per wave", "S12 = S4, plain stores, 1.5 MB window", "S13 = S4, y0 only", "S14 = S4, even slices store",
                            "S15 = S4, loads of y, no stores", "S16 = S4, y fine-grained", "S17 = S4, y uncached", "S18 = S6, pieces of y transposed",
                            "S19 = S6, pieces hashed within 1024", "S20 = S6, neighbours swapped", "S21 = S6, stores sc0", "S22 = S6, stores sc1",
                            "S23 = S6, stores sc0 sc1", "S24 = S6, stores sc0 sc1 nt", "S25 = S6, y behind the slice's values",
                            "S26 = S2 + stores at the wave's START", "S27 = S2 + stores mid-way", "S28 = S4 as PERSISTENT waves, XCD queues",
                            "S29 = S28, queue by blockIdx % 8", "S30 = S28 capped at 6 workgroups per CU", "", "", "S33 = S4 + p.Ap epilogue (barrier)",
                            "S34 = S4 + p.Ap epilogue, no barrier", "S35 = S4, bases by per-lane loads", "S36 = S4, x window read lane-contiguous"};
    if (argc > 2 && strstr(argv[2], "sweep")) {
        // where y lies: S4 timed with y carved out of the start of 28 spacer blocks of 8 GB, allocated one after the other
        printf("S2 (no stores): %.4f ms;  S4 with the y of the ordinary allocation order: %.4f ms\n",
               run<2>(nslices, nb, vals, cw, base, x, y, sink, 20), run<4>(nslices, nb, vals, cw, base, x, y, sink, 20));
        std::vector<double *> sp;
        for (int i = 0; i < 28; i++) {
            double *q = nullptr;
            if (hipMalloc(&q, (size_t)8 << 30) != hipSuccess) { (void)hipGetLastError(); break; }
            sp.push_back(q);
        }
        for (size_t i = 0; i < sp.size(); i++) {
            const float a = run<4>(nslices, nb, vals, cw, base, x, sp[i], sink, 20);
            const float b = run<4>(nslices, nb, vals, cw, base, x, sp[i] + ((size_t)4 << 27), sink, 20);   // 4 GB further in
            printf("  y in spacer %2zu (%p): %.4f ms   4 GB further in: %.4f ms\n", i, (void *)sp[i], a, b);
        }
        // the column words moved (y in the best place found above, x where it was)
        {
            size_t jb = 0; float tbest = 1e30f;
            for (size_t i = 0; i < sp.size(); i++) { const float t = run<4>(nslices, nb, vals, cw, base, x, sp[i], sink, 10); if (t < tbest) { tbest = t; jb = i; } }
            printf("  best y: spacer %zu (%.4f ms); column words copied into other spacers:\n", jb, tbest);
            for (size_t i = 0; i < sp.size(); i += 2) {
                if (i == jb) continue;
                uint32_t *cw2 = (uint32_t *)(sp[i] + ((size_t)1 << 27));
                (void)hipMemcpy(cw2, cw, (size_t)nslices * 14 * 64 * 4, hipMemcpyDeviceToDevice);
                printf("    column words in spacer %2zu: %.4f ms\n", i, run<4>(nslices, nb, vals, cw2, base, x, sp[jb], sink, 20));
            }
            printf("    column words where they were: %.4f ms\n", run<4>(nslices, nb, vals, cw, base, x, sp[jb], sink, 20));
        }
        // and x (the gather vector) moved instead, y where it was
        for (size_t i = 0; i < sp.size(); i += 4) {
            (void)hipMemset(sp[i], 0, (size_t)(nslices * 64LL + 64) * 3 * 8);
            printf("  x in spacer %2zu: %.4f ms (S4), %.4f ms (S2)\n", i, run<4>(nslices, nb, vals, cw, base, sp[i], y, sink, 20),
                   run<2>(nslices, nb, vals, cw, base, sp[i], y, sink, 20));
        }
        return 0;
    }
    if (argc > 2 && strstr(argv[2], "data")) {
        // does the CONTENT of the value stream matter?  zeros (what a fresh block and this lab's other modes hold) against noise
        printf("values all zero:      S0 %.4f  S2 %.4f  S4 %.4f ms\n", run<0>(nslices, nb, vals, cw, base, x, y, sink, 40),
               run<2>(nslices, nb, vals, cw, base, x, y, sink, 40), run<4>(nslices, nb, vals, cw, base, x, y, sink, 40));
        {
            std::vector<double> h((size_t)1 << 24);
            unsigned long long z = 88172645463325252ull;
            for (double &d : h) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; d = (double)(long long)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5; }
            for (size_t o = 0; o < nv; o += h.size()) CK(hipMemcpy(vals + o, h.data(), std::min(h.size(), nv - o) * 8, hipMemcpyHostToDevice));
        }
        printf("values random fp64:   S0 %.4f  S2 %.4f  S4 %.4f ms\n", run<0>(nslices, nb, vals, cw, base, x, y, sink, 40),
               run<2>(nslices, nb, vals, cw, base, x, y, sink, 40), run<4>(nslices, nb, vals, cw, base, x, y, sink, 40));
        CK(hipMemset(vals, 0, nv * 8));
        printf("values all zero again: S0 %.4f  S2 %.4f  S4 %.4f ms\n", run<0>(nslices, nb, vals, cw, base, x, y, sink, 40),
               run<2>(nslices, nb, vals, cw, base, x, y, sink, 40), run<4>(nslices, nb, vals, cw, base, x, y, sink, 40));
        return 0;
    }
    // argv[2]: comma-separated steps (default: all), argv[3]: timed launches per step (default 40)
    bool want[37];
    for (int i = 0; i < 37; i++) want[i] = argc <= 2 && i != 31 && i != 32;
    if (argc > 2) for (const char *q = argv[2]; *q;) { want[atoi(q) % 37] = true; while (*q && *q != ',') q++; if (*q) q++; }
    const int reps = argc > 3 ? atoi(argv[3]) : 40;
    for (int round = 0; round < (argc > 3 ? 1 : 2); round++) {
        float t[37] = {0};
#define RUN(S) if (want[S]) t[S] = run<S>(nslices, nb, vals, cw, base, x, y, sink, reps);
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15)
        RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) RUN(23) RUN(24) RUN(25) RUN(26) RUN(27) RUN(33) RUN(34) RUN(35) RUN(36)
#undef RUN
        for (int st = 31; st <= 32; st++)
            if (argc <= 2 || strstr(argv[2], st == 31 ? "31" : "32")) {
                const float ms = st == 31 ? run<31>(nslices, nb, vals, cw, base, x, y, sink, reps) : run<32>(nslices, nb, vals, cw, base, x, y, sink, reps);
                std::vector<long long> tm((size_t)nslices * 2);
                CK(hipMemcpy(tm.data(), (char *)sink + 64, tm.size() * 8, hipMemcpyDeviceToHost));
                double a = 0, b = 0; long long mx = 0;
                for (int i = 0; i < nslices; i++) { a += tm[2 * i]; b += tm[2 * i + 1]; mx = std::max(mx, tm[2 * i]); }
                printf("  S%d: %.4f ms; per wave: %s -> acknowledged %.2f us on average (max %.2f), whole slice %.2f us\n", st, ms,
                       st == 31 ? "stores" : "loads of y", a / nslices * 0.01, mx * 0.01, b / nslices * 0.01);
            }
        if (want[28]) t[28] = run_persist<0>(nslices, nb, vals, cw, base, x, y, cnt, reps, 0);
        if (want[29]) t[29] = run_persist<1>(nslices, nb, vals, cw, base, x, y, cnt, reps, 0);
        if (want[30]) t[30] = run_persist<0>(nslices, nb, vals, cw, base, x, y, cnt, reps, 6);
        if (want[16]) t[16] = yf ? run<4>(nslices, nb, vals, cw, base, x, yf, sink, reps) : 0.f;
        if (want[17]) t[17] = yu ? run<4>(nslices, nb, vals, cw, base, x, yu, sink, reps) : 0.f;
        for (int i = 0; i < 37; i++)
            if (want[i] && i != 31 && i != 32)
                printf("  %-34s %.4f ms   values / t = %.0f GB/s, all bytes / t = %.0f GB/s\n", names[i], t[i], gb_vals / (t[i] * 1e-3),
                       (i == 0 ? gb_vals : i == 1 ? gb_vals + nslices * 14 * 256 / 1e9 : gb_all) / (t[i] * 1e-3));
    }
    return 0;
}
