// cg.hip -- Conjugate Gradient on gfx950: replaces SolverFunctions.LinearSolver_CG
// (SolverFunctions.cs:270-330), i.e. alglib.lincgcreate / lincgsetcond /
// lincgsolvesparse / lincgresults of alglib.net 3.16.0 (not vendored in the
// reference; its published algorithm is restated here):
//   * diagonal preconditioner applied as a symmetric scaling  A^ = S K S,
//     s_i = 1/sqrt(K_ii) (1 when K_ii <= 0), b^ = S b, result U = S x^;
//   * x0 = 0; stop when ||r^|| <= EpsF ||b^|| (type 1), after MaxIts > 0 iterations
//     (type 5), when the merit function x'Ax - 2b'x stops decreasing (type 7, previous
//     point returned), p'Ap <= 0 (type -5) or non-finite numbers (type -4);
//   * every 10th iteration the residual is recomputed as b^ - A^ x^ (extra SpMV).
//
// All kernels here are HBM-bound streaming kernels.  The scaling is folded into the
// matrix once per matrix, so an iteration is
//   SpMV (+ fused p.Ap)                      reads the matrix once
//   step  : x' = x + a p, r -= a v, r.r, merit   (one pass, 5 reads 2 writes)
//   update: p = r + b p                          (one pass, 2 reads 1 write)
// plus two 1-block reductions of per-block partial sums (fixed order => the whole solve is
// bit-reproducible).  alpha, beta and every stopping decision live in device memory; the
// host only enqueues iterations and polls a status word every CHUNK iterations, so
// there is no host synchronisation inside an iteration.
#include <chrono>
#include <cmath>
#include <thread>

#include "internal.h"
#include "p2p_device.h"

// lab switches (tools/lib_lab.sh): non-temporal policy of the vector traffic
#ifndef STAN_VEC_NT
#define STAN_VEC_NT 1
#endif
#ifndef STAN_Y_NT
#define STAN_Y_NT 1
#endif
// round 2 (tools/fold_ab.py, profiles/r02/fold_ab_incg_*.txt): the in-CG penalty of the SpMV is the
// REWRITING of its gather vector between two products, nothing else (a k_step pass in between costs
// nothing).  Rewritten by plain stores the following product ran 1.5 / 1.6 / 2.7 / 3.9 / 11 %
// slower than back to back on five boxes; by non-temporal stores (no load of the line before)
// 1.0 / 1.1 / 1.1 / 1.0 %; non-temporal stores followed by one streaming read of the vector: 0 %
// (but that read costs what it saves).  Agent- / system-scope (write-through) stores: like plain.
// k_update is a read-modify-write of p: a non-temporal store to a line its own plain load has just
// brought into L2 changes nothing, non-temporal loads AND stores recover a part (0.2 % where the
// penalty is 1.6 %).  That is what STAN_OPT_VEC_STORE_NT selects; it never costs anything.
// Writing the new p into the OTHER of two buffers (no line of it in any cache) was tried as well:
// +1.2 % against +1.3 % for the in-place form with non-temporal loads and stores: not built.
// (STAN_OPT_VEC_STORE_NT: bit 0 = k_update stores p non-temporally, bit 1 = k_step stores r so.)
#ifndef STAN_R_NT
#define STAN_R_NT 1   // k_refresh: the new residual stored non-temporally
#endif

namespace {

constexpr int VEC_BLOCKS = 2048;  // grid of the streaming vector kernels (8 blocks per CU)
constexpr int VEC_T = 256;
constexpr int CHUNK = 32;         // iterations enqueued between two status polls
constexpr int CHUNK_DIST = 8;     // ... of a sharded loop

// device scalar slots (double)
enum { S_BNORM = 0, S_VMV = 1, S_R2NEW = 2, S_MERIT = 3, S_RHO0 = 4, S_RHO1 = 5, S_PMF0 = 6,
       S_PMF1 = 7, S_R2OUT = 8,
       // single-reduction loop: the three sums of one iteration are contiguous (ONE all-reduce)
       S_SR_GAMMA = 9, S_SR_DELTA = 10, S_SR_MERIT = 11, S_SR_GP0 = 12, S_SR_GP1 = 13, S_SR_AP0 = 14,
       S_SR_AP1 = 15, S_NSCAL = 24 };
// device status slots (int64)
enum { T_ITER_A = 0, T_ITER_B = 1, T_TYPE = 2, T_ITERS = 3, T_XSEL = 4, T_NSTAT = 8 };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
// block sum (256 threads), valid in thread 0; fixed combination order
__device__ __forceinline__ double block_sum(double v, double *sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// ---- reductions folded into their producers ("last block done") --------------------------------
// Every block of a producing kernel leaves its partial sum(s) in `partial`, then takes a ticket;
// the block that draws the last ticket adds ALL partials in a fixed order (independent of which
// block that is: the result is bit-reproducible) and writes the scalar.  That removes the two
// one-block k_reduce launches per iteration from the stream (2 x (4.5 us + a kernel boundary) at
// 148^3; more where it matters: the sharded loop, whose per-rank kernels are 8 x shorter).
// Hand-off across XCDs (their L2s are not coherent, MI355X_MICROARCH.md "inter-workgroup
// visibility", first row of the table of measured forms): the partial is an agent-scope store
// (sc1, write-through), the storing lane waits for it (vmcnt(0)) before its agent-scope add to
// the one unsharded counter, the block whose add returned the last ticket reads every partial
// with agent-scope (sc1) loads after a workgroup barrier behind that add.
// STAN_OPT_CG_FOLD_REDUCE = 0 restores the separate k_reduce launches (same order: same bits).
__device__ __forceinline__ void st_agent(double *p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// r[j] = sum_i partial[i*NV + j] over np blocks by one 256-thread block, all NV sums in ONE pass
// over the partials (their loads overlap), fixed order; valid in thread 0
template <int NV>
__device__ __forceinline__ void sum_partials(const double *partial, int np, double *sh, double r[NV]) {
    constexpr int W = 16 / NV;   // loads in flight per thread: the last block's latency adds to the kernel
    double a[NV][W];
#pragma unroll
    for (int j = 0; j < NV; j++)
#pragma unroll
        for (int q = 0; q < W; q++) a[j][q] = 0;
    int i = threadIdx.x;
    for (; i + (W - 1) * 256 < np; i += W * 256) {
#pragma unroll
        for (int q = 0; q < W; q++)
#pragma unroll
            for (int j = 0; j < NV; j++) a[j][q] += ld_agent(partial + (int64_t)(i + q * 256) * NV + j);
    }
    for (; i < np; i += 256)
#pragma unroll
        for (int j = 0; j < NV; j++) a[j][0] += ld_agent(partial + (int64_t)i * NV + j);
#pragma unroll
    for (int j = 0; j < NV; j++) {
#pragma unroll
        for (int w = W / 2; w > 0; w >>= 1)   // fixed pairwise tree
#pragma unroll
            for (int q = 0; q < w; q++) a[j][q] += a[j][q + w];
        r[j] = block_sum(a[j][0], sh);
    }
}
// Tickets are two-level: block b first counts itself into sub-counter b % FOLD_SUB (a 128-B line
// of its own), the last arrival of a sub-counter counts that sub-counter into the top counter, the
// last arrival there finishes.  One flat counter cost k_step +10 us (rocprofv3, 148^3): its 2048
// blocks end together and 2048 adds to ONE address are served one after the other at the memory
// side; with 32 sub-counters the longest queue is 64.
constexpr int FOLD_SUB = 32;
constexpr int FOLD_LINE = 16;                               // uint64 per 128-B line
constexpr int FOLD_WORDS = (1 + FOLD_SUB) * FOLD_LINE;      // one counter set: top + sub-counters
struct fold_args {
    unsigned long long *counter;  // counter set (zero between kernels); nullptr: no fold
    unsigned nblocks;             // tickets this launch hands out (its grid size)
    int np;                       // partials to add (>= nblocks: earlier launches may have left some)
    double *out;                  // [NV] results
    p2p_out po;                   // sharded, peer to peer: the sums go to every rank's mailbox instead (p2p_device.h)
};
constexpr p2p_out NO_P2P = {nullptr, 0, 0, 0};
constexpr fold_args NO_FOLD = {nullptr, 0, 0, nullptr, NO_P2P};
// the finished sums r[0..NV) (valid in thread 0) to where the consumer will look for them
template <int NV>
__device__ __forceinline__ void publish_sums(double *out, const p2p_out &po, const double r[NV], double *sh) {
    if (po.pp) { p2p_publish<NV>(po, r, sh); return; }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < NV; j++) out[j] = r[j];  // read by the NEXT kernel: a plain store will do
    }
}
// Thread 0 of every block calls this after storing its partials with st_agent(); true (in every
// thread) for the block that arrived last.  `sh_last` is one int of LDS.
__device__ __forceinline__ bool fold_arrive(const fold_args &f, int *sh_last) {
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the partial has left this CU
        const unsigned sub = blockIdx.x % FOLD_SUB;
        const unsigned in_sub = (f.nblocks - sub + FOLD_SUB - 1) / FOLD_SUB;     // blocks b with b % SUB == sub
        const unsigned nsub = f.nblocks < (unsigned)FOLD_SUB ? f.nblocks : (unsigned)FOLD_SUB;
        int last = 0;
        unsigned long long t = __hip_atomic_fetch_add(f.counter + (1 + sub) * FOLD_LINE, 1ULL, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned long long)in_sub - 1) {
            t = __hip_atomic_fetch_add(f.counter, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = t == (unsigned long long)nsub - 1;
        }
        *sh_last = last;
    }
    __syncthreads();
    return *sh_last != 0;
}
template <int NV>
__device__ __forceinline__ void fold_finish(const fold_args &f, const double *partial, double *sh) {
    double r[NV];
    sum_partials<NV>(partial, f.np, sh, r);
    publish_sums<NV>(f.out, f.po, r, sh);
    // every ticket of this launch has been drawn: clear the set for the next one
    if (threadIdx.x <= FOLD_SUB)
        __hip_atomic_store(f.counter + threadIdx.x * FOLD_LINE, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A producing kernel that returns without doing its work (the solve has stopped; every rank takes the same
// decision) still owes its peers the arrival count of its reduction: their streams wait for it.
__device__ __forceinline__ void fold_skip(const fold_args &f) {
    if (f.counter && f.po.pp && f.po.signal && blockIdx.x == 0 && (int)threadIdx.x < f.po.pp->n)
        __hip_atomic_fetch_add(f.po.pp->sig_red[threadIdx.x][f.po.slot], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ bool stopped(const int64_t *st, int64_t k) {
    return st[T_ITER_A] < k || st[T_ITER_B] < k;
}

// ---- setup kernels ---------------------------------------------------------------------------
// s_i = 1/sqrt(K_ii) if K_ii > 0 else 1 (lincgsolvesparse); one lane per block row
__global__ void k_diag_scale(int64_t nloc, const int32_t *rowlen, const int32_t *posof, const int32_t *slot_ptr,
                             const int32_t *cols, const double *vals, double *s) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nloc) return;
    const int64_t pos = posof[row];   // SELL-C-sigma: where the row sits in the sliced layout
    const int64_t slice = pos >> 6;
    const int lane = (int)(pos & 63);
    const int32_t k0 = slot_ptr[slice];
    double d0 = 0, d1 = 0, d2 = 0;
    for (int k = 0; k < rowlen[row]; k++)
        if (cols[((int64_t)k0 + k) * 64 + lane] == (int32_t)row) {
            const double *v = vals + ((int64_t)k0 + k) * 9 * 64 + lane;
            d0 = v[0 * 64]; d1 = v[4 * 64]; d2 = v[8 * 64];
            break;
        }
    s[3 * row + 0] = d0 > 0 ? 1.0 / sqrt(d0) : 1.0;
    s[3 * row + 1] = d1 > 0 ? 1.0 / sqrt(d1) : 1.0;
    s[3 * row + 2] = d2 > 0 ? 1.0 / sqrt(d2) : 1.0;
}

// vals[slot][3m+n][lane] *= s[3 row + m] * s[3 col + n]   (inverse=1: divide)
__global__ void __launch_bounds__(256)
k_scale_matrix(int32_t nslices, const int32_t *slot_ptr, const int32_t *rowof, const int32_t *cols, double *vals,
               const double *s, int inverse) {
    const int lane = threadIdx.x & 63;
    const int64_t slice = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slice >= nslices) return;
    const int64_t row = rowof[slice * 64 + lane];
    double sr[3] = {s[3 * row], s[3 * row + 1], s[3 * row + 2]};  // s is padded to slices
    if (inverse) { sr[0] = 1.0 / sr[0]; sr[1] = 1.0 / sr[1]; sr[2] = 1.0 / sr[2]; }
    for (int32_t k = slot_ptr[slice]; k < slot_ptr[slice + 1]; k++) {
        const int32_t c = cols[(int64_t)k * 64 + lane];
        double sc[3] = {s[3 * (int64_t)c], s[3 * (int64_t)c + 1], s[3 * (int64_t)c + 2]};
        if (inverse) { sc[0] = 1.0 / sc[0]; sc[1] = 1.0 / sc[1]; sc[2] = 1.0 / sc[2]; }
        double *v = vals + (int64_t)k * 9 * 64 + lane;
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int n = 0; n < 3; n++) v[(3 * m + n) * 64] *= sr[m] * sc[n];
    }
}

__global__ void k_to_fp32(const double *in, float *out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = (float)in[i];
}

constexpr double FX48_ONE = 70368744177664.0;                          // 2^46
constexpr double FX48_INV = 1.0 / 70368744177664.0;                    // 2^-46
constexpr double FX48_BIAS = 4503599627370496.0 + 140737488355328.0;   // 2^52 + 2^47

// scaled fp64 values -> FIXED-48 stream (see vstream<uint32_t>); *bad counts the entries
// with |a| >= 2 (not representable: the matrix was not SPD-scalable)
__global__ void __launch_bounds__(256)
k_to_fx48(int64_t nslots, const double *vals, uint32_t *out, unsigned long long *bad) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t slot = t >> 6;
    const int lane = (int)(t & 63);
    if (slot >= nslots) return;
    const double *v = vals + slot * 9 * 64 + lane;
    uint32_t *o = out + slot * 14 * 64 + lane;
    uint32_t hi[10];
    int nbad = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const double a = v[j * 64] * FX48_ONE;
        long long q = 0;
        if (!(fabs(a) < 140737488355328.0)) nbad++;  // also catches NaN
        else q = __double2ll_rn(a);
        if (q >= 140737488355328LL) { q = 0; nbad++; }
        const unsigned long long u = (unsigned long long)(q + 140737488355328LL);
        o[j * 64] = (uint32_t)u;
        hi[j] = (uint32_t)(u >> 32);
    }
    hi[9] = 0;
#pragma unroll
    for (int m = 0; m < 5; m++) o[(9 + m) * 64] = hi[2 * m] | (hi[2 * m + 1] << 16);
    if (nbad) atomicAdd(bad, (unsigned long long)nbad);
}

// packed column stream (struct colstream): one wavefront per slice.  Mode of a slice (ok[slice]):
//   1  every slot's 64 columns (padding entries = the row's own column included) lie within 2^16 of the slot's smallest:
//      one base per slot (round 2);
//   2  (round 4) the slice mixes rows of different length -- the k-th neighbour of a short row (a node on the surface
//      of the mesh) plays another part than the k-th neighbour of its 27-neighbour slice mates and, once a breadth-first
//      level is wider than 2^16 rows (200^3: 120 k), lies further away than an offset reaches.  Two bases per slot: A
//      for the rows of the slice's full width, B for the shorter ones (cmask[slice]: one bit per lane); a padding
//      entry (zero values) takes offset 0 from its class's base.  63.9 % -> 99.8 % of the slots at 200^3 / 400^3,
//      98.4 % -> 99.9 % at 148^3 (tools: /profiles/r04/packed_columns_two_bases.txt);
//   0  neither: the slice keeps the int32 stream.
// rowof == nullptr (the folded copy's stream, whose lanes carry foreign pieces): modes 0 / 1 only.
__global__ void __launch_bounds__(256)
k_pack_cols(int32_t nslices, int64_t nloc, const int32_t *slot_ptr, const int32_t *cols, const int32_t *rowof, const int32_t *rowlen,
            const int32_t *pair_ptr, uint32_t *packed, int32_t *base, int32_t *base2, unsigned long long *cmask, uint8_t *ok) {
    const int lane = threadIdx.x & 63;
    const int64_t slice = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slice >= nslices) return;
    const int32_t k0 = slot_ptr[slice], k1 = slot_ptr[slice + 1];
    int32_t len = k1 - k0;   // without row lengths every entry counts as live and every lane as class A
    if (rowof) {
        const int64_t row = rowof[slice * 64 + lane];
        len = row < nloc ? rowlen[row] : 0;
    }
    const bool cls_b = len < k1 - k0;
    const int32_t BIG = 0x7fffffff;
    auto wmin = [](int32_t v) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v = min(v, __shfl_xor(v, d, 64));
        return v;
    };
    auto wmax = [](int32_t v) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
        return v;
    };
    bool fits1 = true, fits2 = rowof != nullptr;
    for (int32_t k = k0; k < k1; k++) {
        const int32_t c = cols[(int64_t)k * 64 + lane];
        const bool live = k - k0 < len;
        fits1 = fits1 && (wmax(c) - wmin(c)) < 65536;
        if (fits2) {
            const int32_t mna = wmin(live && !cls_b ? c : BIG), mxa = wmax(live && !cls_b ? c : -1);
            const int32_t mnb = wmin(live && cls_b ? c : BIG), mxb = wmax(live && cls_b ? c : -1);
            fits2 = (mxa < 0 || mxa - mna < 65536) && (mxb < 0 || mxb - mnb < 65536);
        }
    }
    const int mode = fits1 ? 1 : fits2 ? 2 : 0;
    uint32_t *out = packed + (int64_t)pair_ptr[slice] * 64 + lane;
    uint32_t lo = 0;
    for (int32_t k = k0; k < k1; k++) {
        const int32_t c = cols[(int64_t)k * 64 + lane];
        uint32_t dlt;
        if (mode == 2) {
            const bool live = k - k0 < len;
            const int32_t mna = wmin(live && !cls_b ? c : BIG);   // (never BIG: the rows of full width are live in every slot)
            int32_t mnb = wmin(live && cls_b ? c : BIG);
            if (mnb == BIG) mnb = mna;                            // no short row reaches this slot: its padding points at A's base
            if (lane == 0) { base[k] = mna; base2[k] = mnb; }
            dlt = live ? (uint32_t)(c - (cls_b ? mnb : mna)) & 0xffffu : 0u;
        } else {
            const int32_t mn = wmin(c);
            if (lane == 0) { base[k] = mn; base2[k] = mn; }
            dlt = (uint32_t)(c - mn) & 0xffffu;
        }
        if (((k - k0) & 1) == 0) lo = dlt;
        else { *out = lo | (dlt << 16); out += 64; }
    }
    if ((k1 - k0) & 1) *out = lo;
    const unsigned long long mb = __ballot(cls_b);
    if (lane == 0) {
        ok[slice] = (uint8_t)mode;
        cmask[slice] = mode == 2 ? mb : 0ULL;
    }
}
__global__ void k_pair_counts(int32_t nslices, const int32_t *slot_ptr, int32_t *cnt) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nslices) cnt[s] = (slot_ptr[s + 1] - slot_ptr[s] + 1) >> 1;
}
__global__ void k_count_ok(int32_t nslices, const uint8_t *ok, const int32_t *slot_ptr, unsigned long long *out) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nslices && ok[s]) atomicAdd(out, (unsigned long long)(slot_ptr[s + 1] - slot_ptr[s]));           // packed slots
    if (s < nslices && ok[s] == 2) atomicAdd(out + 1, (unsigned long long)(slot_ptr[s + 1] - slot_ptr[s]));  // ... with two bases
}

// b^[i] = s_i * F[d - red[d]] on free DOFs, 0 on fixed ones; also x0 = 0, r = p = b^ and
// partial sums of b^.b^ (x0 = 0 => r0 = b^, merit0 = 0).
__global__ void __launch_bounds__(VEC_T)
k_init(int64_t n3, int64_t dof0, const int32_t *red, const double *F, const double *s,
       double *bh, double *x0, double *r, double *p, double *partial, fold_args fold) {
    __shared__ double sh[4];
    __shared__ int sh_last;
    double acc = 0;
    const int64_t stride = (int64_t)gridDim.x * VEC_T;
    for (int64_t i = (int64_t)blockIdx.x * VEC_T + threadIdx.x; i < n3; i += stride) {
        const int32_t rd = red[dof0 + i];
        const double b = rd == -1 ? 0.0 : s[i] * F[dof0 + i - rd];
        bh[i] = b;
        x0[i] = 0.0;
        r[i] = b;
        p[i] = b;
        acc += b * b;
    }
    const double t = block_sum(acc, sh);
    if (threadIdx.x == 0) st_agent(partial + blockIdx.x, t);
    if (fold.counter && fold_arrive(fold, &sh_last)) fold_finish<1>(fold, partial, sh);
}

// out[j] = sum_i partial[i*nv + j]: the unfolded form of the reduction (one 256-thread block, the
// same summation order as fold_finish: both paths give the same bits)
// st != nullptr (the loop's launches, peer to peer): once the solve has stopped -- every rank takes the same decision, but
// not at the same moment -- this launch only COUNTS, like fold_skip: a rank that free-runs through the iterations enqueued
// behind the stop (STAN_P2P_WAIT_MODE=2: stopped consumers do not poll) must not store stale partials into a mailbox slot
// that a slower peer has not read yet (slot R + RING aliases slot R).
template <int NV>
__global__ void __launch_bounds__(256)
k_reduce(const double *partial, int np, double *out, p2p_out po, const int64_t *st, int64_t k) {
    __shared__ double sh[4];
    if (st && po.pp && stopped(st, k)) {
        if (po.signal && (int)threadIdx.x < po.pp->n)
            __hip_atomic_fetch_add(po.pp->sig_red[threadIdx.x][po.slot], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    double r[NV];
    sum_partials<NV>(partial, np, sh, r);   // np == 0 (a rank that owns no rows): zeros
    publish_sums<NV>(out, po, r, sh);
}

// after the b^.b^ reduction: bnorm, first residual test, rho, prevmf
__global__ void __launch_bounds__(64) k_init_scalars(double *sc, int64_t *st, double epsf, red_src rs) {
    __shared__ double sh[4];
    double t[1];
    red_get<1>(sc + S_VMV, rs, t, sh);   // b^.b^ (all ranks' partials when sharded peer to peer)
    if (threadIdx.x != 0) return;
    const double r2 = t[0];
    sc[S_BNORM] = sqrt(r2);
    sc[S_RHO0] = r2; sc[S_RHO1] = r2;  // iteration 1 reads slot 1
    sc[S_PMF0] = 0.0; sc[S_PMF1] = 0.0;
    sc[S_R2OUT] = r2;
    st[T_ITER_A] = 0x7fffffffffffffffLL;
    st[T_ITER_B] = 0x7fffffffffffffffLL;
    st[T_TYPE] = 0; st[T_ITERS] = 0; st[T_XSEL] = 0;
    if (!isfinite(r2)) { st[T_TYPE] = -4; st[T_ITER_A] = 0; }
    else if (sqrt(r2) <= epsf * sqrt(r2)) { st[T_TYPE] = 1; st[T_ITER_A] = 0; }
}

// ---- SpMV --------------------------------------------------------------------------------------
// y = A x over BSELL-64.  One wavefront per slice, one lane per block row: every load of
// the value stream is a contiguous 512-B (fp64) / 256-B (fp32) wave access; x is gathered
// (24 B per block, L2 / Infinity-Cache resident: neighbouring rows share columns).
// DOT = 1: also the per-block partial of x_own . y  (p.Ap of the CG); DOT = 2: the partials of
// x_own . x_own and x_own . y (r.r and r.Ar of the single-reduction CG); the block that finishes
// last adds the partials up (fold_args).
// VAR selects the kernel variant.  The product library carries three:
//   0   plain loads, identity workgroup mapping (reference point of the A/B runs)
//   9   non-temporal loads for the once-read matrix stream (keeps x in L2 / MALL) + XCD-chunked
//       workgroup mapping (below)                                   -- default for fp64 / fp32
//   12  = 9 with the block loop unrolled by 4                       -- default for FIXED-48
// A lab build (make lab: -DSTAN_LAB, build_lab/libstan_hip_lab.so, never shipped) adds the other
// A/B variants of round 1 (1-8, 10, 11, 13; 8 is a timing-only kernel whose results are wrong).
// NT must be a compile-time choice: with a run-time flag, `nt ? __builtin_nontemporal_load(p) : *p`
// is two loads of one address that the optimiser merges into ONE plain load inside this helper,
// before it is inlined anywhere -- the non-temporal hint never reached the ISA (no `nt` bit on any
// global_load of the first builds; found by reading the disassembly).
template <bool NT, typename T>
__device__ __forceinline__ T ld_stream(const T *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

// Value streams of the matrix.  double / float: vals[slot][9][64].
// uint32_t = FIXED-48 (STAN_PREC_FIXED48): after the Jacobi scaling every entry of an SPD matrix
// satisfies |a_ij| <= sqrt(a_ii a_jj) = 1, so no exponent is needed: q = rint(a * 2^46) as a
// signed 48-bit integer (absolute error <= 2^-47 = 7.1e-15 of the unit diagonal), stored
// offset-binary u = q + 2^47 as vals48[slot][14][64] dwords: rows 0..8 the low 32 bits of the
// nine entries, rows 9..13 the high 16 bits packed two per dword.  60 B per block instead of
// 76 B.  Decoding is one integer op and one exact fp64 subtraction per entry: the bits
// 0x43300000'00000000 | u are the double 2^52 + u.  The 2^-46 is applied to the three
// gathered x values instead of the nine entries (powers of two commute with rounding), so
// the arithmetic is the fp64 product with the quantised matrix.
template <typename VT> struct vstream { static constexpr int STRIDE = 9 * 64; static constexpr bool FX = false; };
template <> struct vstream<uint32_t> { static constexpr int STRIDE = 14 * 64; static constexpr bool FX = true; };

template <bool NT, typename VT>
__device__ __forceinline__ void load9(const VT *vp, double a[9]) {
    if constexpr (vstream<VT>::FX) {
        uint32_t lo[9], hw[5];
#pragma unroll
        for (int j = 0; j < 9; j++) lo[j] = ld_stream<NT>(vp + j * 64);
#pragma unroll
        for (int m = 0; m < 5; m++) hw[m] = ld_stream<NT>(vp + (9 + m) * 64);
#pragma unroll
        for (int j = 0; j < 9; j++) {
            const uint32_t h = (j & 1) ? (hw[j >> 1] >> 16) : (hw[j >> 1] & 0xffffu);
            a[j] = __hiloint2double((int)(0x43300000u | h), (int)lo[j]) - FX48_BIAS;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 9; j++) a[j] = (double)ld_stream<NT>(vp + j * 64);
    }
}

// Packed column stream (round 2): the block-column indices of a slot are, in reference (BFS) DOF
// order, within a few thousand of each other across the 64 rows of a slice, so a slot stores its
// smallest column once (a wave-uniform scalar) and every lane a 16-bit offset, two slots per
// dword: 2 B per block instead of 4 (74 B instead of 76 with fp64 values: -2.6 % of the SpMV's
// bytes, lossless, same products in the same order -> same bits).  Slices whose offsets do not fit
// (the ragged last slice, rows coupling owned and halo columns far apart) keep the int32 stream.
struct colstream {
    const uint32_t *packed;   // [pair][64]: offset of slot 2j in the low half, of slot 2j+1 in the high half
    const int32_t *base;      // [slot] smallest column of the slot
    const int32_t *pair_ptr;  // [nslices + 1] first pair of every slice
    const uint8_t *ok;        // [nslices] 0 = int32 columns, 1 = packed, one base per slot, 2 = packed, two bases (k_pack_cols)
    const int32_t *base2;     // [slot] mode 2: the base of the slice's SHORTER rows (base: of its full-width rows)
    const unsigned long long *cmask;   // [nslices] mode 2: bit l = lane l holds a shorter row
};
constexpr colstream NO_COLSTREAM = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
// the arrays behind `base` (one allocation, stan_pack_columns): [nslots] base, [nslots] base2, [nslices] cmask
inline colstream make_colstream(const uint32_t *packed, const int32_t *base, const int32_t *pair_ptr, const uint8_t *ok, int64_t nslots) {
    const int64_t n = nslots > 0 ? nslots : 1;
    return colstream{packed, base, pair_ptr, ok, base + n, (const unsigned long long *)(base + 2 * n)};
}

#define STAN_SPMV_BLOCK(C, VP)                                                        \
    {                                                                                 \
        double a[9];                                                                  \
        load9<NT, VT>(VP, a);                                                         \
        double x0 = x[3 * (C)], x1 = x[3 * (C) + 1], x2 = x[3 * (C) + 2];             \
        if (vstream<VT>::FX) { x0 *= FX48_INV; x1 *= FX48_INV; x2 *= FX48_INV; }      \
        y0 += a[0] * x0 + a[1] * x1 + a[2] * x2;                                      \
        y1 += a[3] * x0 + a[4] * x1 + a[5] * x2;                                      \
        y2 += a[6] * x0 + a[7] * x1 + a[8] * x2;                                      \
    }

template <typename VT, int DOT, int VAR>
// lab 17 / 18: capped at 68 / 62 VGPRs for 7 / 8 waves per SIMD instead of 78 / 6: in-CG SpMV 1.071 /
// 1.072 ms against 1.032 ms (profiles/r02/fold_ab_incg_n148_box9_register_caps.txt): more waves do not help
__global__ void __launch_bounds__(256, (VAR == 17 ? 7 : VAR == 18 ? 8 : 1))
k_spmv(int32_t nslices, int64_t nloc, const int32_t *__restrict__ slot_ptr, const int32_t *__restrict__ rowof,
       const int32_t *__restrict__ cols, const VT *__restrict__ vals,
       const double *__restrict__ x, double *__restrict__ y, double *partial,
       const int64_t *st, int64_t kiter, const int32_t *__restrict__ slist, int32_t nlist,
       int32_t poff, fold_args fold, colstream cs) {
    __shared__ double sh[4];
    __shared__ int sh_last;
    if (stopped(st, kiter)) { fold_skip(fold); return; }
    constexpr bool NT = ((VAR & 1) != 0 && VAR < 8) || (VAR >= 9 && VAR != 13);  // 13 = 9 without the hint; 14-16 lab
    constexpr bool XCD = (VAR & 2) != 0 && VAR < 8;
    constexpr int UNR = (((VAR & 4) != 0 && VAR < 8) || VAR == 12) ? 4 : 2;
    constexpr int UNR2 = 1;         // the packed-column loop handles two slots per trip (unrolling it further costs 40 VGPRs)
    // VAR 9/10/11: XCD-chunked mapping.  Workgroups go round-robin to the 8 XCDs; here every
    // window of 8*C consecutive workgroups is dealt so that each XCD gets C CONSECUTIVE ones
    // (C = 32 / 8 / 128): an XCD's L2 then holds the x window of one contiguous run of rows
    // while the chip as a whole still sweeps the matrix front to back.
    constexpr int CH = (VAR == 9 || VAR == 12 || VAR >= 13) ? 32 :   /* 17, 18: = 9 with a register cap */ VAR == 10 ? 8 : VAR == 11 ? 128 : 0;  // 12 = 9 + unroll 4
    const int lane = threadIdx.x & 63;
    int64_t bid = blockIdx.x;
    if (XCD) {
        const int64_t g = gridDim.x, cpx = g >> 3, rem = g & 7, xcd = bid & 7;
        bid = xcd * cpx + (xcd < rem ? xcd : rem) + (bid >> 3);
    }
    if (CH > 0) {
        const int64_t win = 8 * CH, grp = bid / win, within = bid - grp * win;
        if ((grp + 1) * win <= (int64_t)gridDim.x)   // the ragged tail keeps the identity mapping
            bid = grp * win + (within & 7) * CH + (within >> 3);
    }
    int64_t slice = bid * 4 + (threadIdx.x >> 6);
    if (slist) slice = slice < nlist ? (int64_t)slist[slice] : (int64_t)nslices;  // interior / boundary list
    double y0 = 0, y1 = 0, y2 = 0;
    const int64_t row = slice < nslices ? (int64_t)rowof[slice * 64 + lane] : nloc;   // SELL-C-sigma: position -> block row
    if (slice < nslices) {
        const int32_t k0 = slot_ptr[slice], k1 = slot_ptr[slice + 1];
        const int32_t *cp = cols + (int64_t)k0 * 64 + lane;
        const VT *vp = vals + (int64_t)k0 * vstream<VT>::STRIDE + lane;
        const int pmode = cs.packed ? (int)cs.ok[slice] : 0;   // wave-uniform: 0 int32 columns, 1 / 2 packed (k_pack_cols)
#ifdef STAN_LAB
#include "lab/spmv_variants_lab.inc"   // lab-only kernel variants (VAR 8, 14-16)
#endif
        if (pmode != 0) {   // one loop for both packed modes: a one-base slice has cmask 0 and base2 = base
            // two bases per slot: the lane's class picks (a select between two scalars)
            const uint32_t *cq = cs.packed + (int64_t)cs.pair_ptr[slice] * 64 + lane;
            const int32_t *bp = cs.base + __builtin_amdgcn_readfirstlane(k0);
            const int32_t *bq = cs.base2 + __builtin_amdgcn_readfirstlane(k0);
            const bool cb = (cs.cmask[slice] >> lane) & 1ull;
            int32_t k = k0;
#pragma unroll UNR2
            for (; k + 1 < k1; k += 2) {
                const uint32_t wd = ld_stream<NT>(cq);
                const int64_t c = (int64_t)(cb ? bq[0] : bp[0]) + (int64_t)(wd & 0xffffu);
                const int64_t c2 = (int64_t)(cb ? bq[1] : bp[1]) + (int64_t)(wd >> 16);
                STAN_SPMV_BLOCK(c, vp)
                STAN_SPMV_BLOCK(c2, vp + vstream<VT>::STRIDE)
                cq += 64;
                bp += 2;
                bq += 2;
                vp += 2 * vstream<VT>::STRIDE;
            }
            if (k < k1) {
                const int64_t c = (int64_t)(cb ? bq[0] : bp[0]) + (int64_t)(ld_stream<NT>(cq) & 0xffffu);
                STAN_SPMV_BLOCK(c, vp)
            }
        } else {
#pragma unroll UNR
            for (int32_t k = k0; k < k1; k++) {
                const int64_t c = ld_stream<NT>(cp);
                STAN_SPMV_BLOCK(c, vp)
                cp += 64;
                vp += vstream<VT>::STRIDE;
            }
        }
        if (row < nloc) {
#if STAN_Y_NT  // A p is read exactly once, by k_step
            if (NT) {
                __builtin_nontemporal_store(y0, y + 3 * row); __builtin_nontemporal_store(y1, y + 3 * row + 1);
                __builtin_nontemporal_store(y2, y + 3 * row + 2);
            } else
#endif
            { y[3 * row] = y0; y[3 * row + 1] = y1; y[3 * row + 2] = y2; }
        }
    }
    if (DOT) {
        double d = 0, e = 0;
        if (slice < nslices && row < nloc) {
            const double x0 = x[3 * row], x1 = x[3 * row + 1], x2 = x[3 * row + 2];
            d = y0 * x0 + y1 * x1 + y2 * x2;
            if (DOT == 2) e = x0 * x0 + x1 * x1 + x2 * x2;
        }
        const double t = block_sum(d, sh);
        if (DOT == 2) {
            const double u = block_sum(e, sh);
            if (threadIdx.x == 0) {
                st_agent(partial + 2 * (int64_t)(blockIdx.x + poff), u);
                st_agent(partial + 2 * (int64_t)(blockIdx.x + poff) + 1, t);
            }
        } else if (threadIdx.x == 0) st_agent(partial + blockIdx.x + poff, t);
        if (fold.counter && fold_arrive(fold, &sh_last)) fold_finish<(DOT == 2 ? 2 : 1)>(fold, partial, sh);
    }
}

// ---- SpMV for SMALL systems -----------------------------------------------------------------------
// Below ~150 k block rows the chip is not filled by one wavefront per slice (46 875 DOF: 245 slices
// on 1024 SIMDs), and a wavefront walking its 27 slots is a chain of dependent memory round trips:
// k_spmv takes 19 us there, two thirds of an iteration (rocprofv3, tools/small_sizes.py).  Here a
// slice belongs to a WORKGROUP: its four wavefronts take every fourth slot, the four partial rows are
// added through LDS in a fixed order (wave 0 + 1 + 2 + 3: deterministic; another order than k_spmv's,
// so the choice between the two depends on the GLOBAL row count only -- every shard of a sharded
// matrix and the unsharded matrix use the same kernel and keep producing the same bits).  Plain
// loads: a matrix of this size stays in the L2s / the memory-side cache from one product to the next.
template <typename VT, int DOT>
__global__ void __launch_bounds__(256)
k_spmv_small(int32_t nslices, int64_t nloc, const int32_t *__restrict__ slot_ptr, const int32_t *__restrict__ rowof,
             const int32_t *__restrict__ cols, const VT *__restrict__ vals,
             const double *__restrict__ x, double *__restrict__ y, double *partial,
             const int64_t *st, int64_t kiter, const int32_t *__restrict__ slist, int32_t nlist,
             int32_t poff, fold_args fold, colstream cs) {
    __shared__ double sh[4];
    __shared__ int sh_last;
    __shared__ double acc[3][4][64];
    if (stopped(st, kiter)) { fold_skip(fold); return; }
    constexpr bool NT = false;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int64_t slice = blockIdx.x;
    if (slist) slice = slice < nlist ? (int64_t)slist[slice] : (int64_t)nslices;
    double y0 = 0, y1 = 0, y2 = 0;
    const int64_t row = slice < nslices ? (int64_t)rowof[slice * 64 + lane] : nloc;   // SELL-C-sigma: position -> block row
    if (slice < nslices) {
        const int32_t k0 = slot_ptr[slice], k1 = slot_ptr[slice + 1];
        const bool packed = cs.packed && cs.ok[slice] == 1;   // (two-base slices read the int32 columns here)
        const int64_t pp = packed ? (int64_t)cs.pair_ptr[slice] : 0;
        for (int32_t k = k0 + w; k < k1; k += 4) {
            int64_t c;
            if (packed) {
                const uint32_t wd = cs.packed[(pp + ((k - k0) >> 1)) * 64 + lane];
                c = (int64_t)cs.base[k] + (int64_t)(((k - k0) & 1) ? (wd >> 16) : (wd & 0xffffu));
            } else
                c = cols[(int64_t)k * 64 + lane];
            const VT *vp = vals + (int64_t)k * vstream<VT>::STRIDE + lane;
            STAN_SPMV_BLOCK(c, vp)
        }
    }
    acc[0][w][lane] = y0; acc[1][w][lane] = y1; acc[2][w][lane] = y2;
    __syncthreads();
    double d = 0, e = 0;
    if (w == 0) {
        y0 = ((acc[0][0][lane] + acc[0][1][lane]) + acc[0][2][lane]) + acc[0][3][lane];
        y1 = ((acc[1][0][lane] + acc[1][1][lane]) + acc[1][2][lane]) + acc[1][3][lane];
        y2 = ((acc[2][0][lane] + acc[2][1][lane]) + acc[2][2][lane]) + acc[2][3][lane];
        if (slice < nslices && row < nloc) {
            y[3 * row] = y0; y[3 * row + 1] = y1; y[3 * row + 2] = y2;
            if (DOT) {
                const double x0 = x[3 * row], x1 = x[3 * row + 1], x2 = x[3 * row + 2];
                d = y0 * x0 + y1 * x1 + y2 * x2;
                if (DOT == 2) e = x0 * x0 + x1 * x1 + x2 * x2;
            }
        }
    }
    if (DOT) {
        const double t = block_sum(d, sh);
        if (DOT == 2) {
            const double u = block_sum(e, sh);
            if (threadIdx.x == 0) {
                st_agent(partial + 2 * (int64_t)(blockIdx.x + poff), u);
                st_agent(partial + 2 * (int64_t)(blockIdx.x + poff) + 1, t);
            }
        } else if (threadIdx.x == 0) st_agent(partial + blockIdx.x + poff, t);
        if (fold.counter && fold_arrive(fold, &sh_last)) fold_finish<(DOT == 2 ? 2 : 1)>(fold, partial, sh);
    }
}

// Two right-hand sides in ONE pass over the matrix: y = A x and y2 = A x2 (+ the x.y partial).
// Used on the residual-refresh iterations: r = b - A(x + a p) = b - (A x + a A p), so the
// refresh needs A x next to the A p every iteration needs -- one matrix stream instead of two.
// 128-VGPR budget (4 waves per SIMD): with the default target the four gathers of a trip (x and x2 of two slots) were
// issued one by one, each behind a wait for earlier data
template <typename VT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 4)))
k_spmv2(int32_t nslices, int64_t nloc, const int32_t *__restrict__ slot_ptr, const int32_t *__restrict__ rowof,
        const int32_t *__restrict__ cols, const VT *__restrict__ vals,
        const double *__restrict__ x, const double *__restrict__ x2, double *__restrict__ y,
        double *__restrict__ y2, double *partial, const int64_t *st, int64_t kiter,
        const int32_t *__restrict__ slist, int32_t nlist, int32_t poff, fold_args fold, colstream cs) {
    __shared__ double sh[4];
    __shared__ int sh_last;
    if (stopped(st, kiter)) { fold_skip(fold); return; }
    const int lane = threadIdx.x & 63;
    int64_t bid = blockIdx.x;
    {   // XCD-chunked workgroup mapping, as in k_spmv (variant 9)
        constexpr int CH = 32;
        const int64_t win = 8 * CH, grp = bid / win, within = bid - grp * win;
        if ((grp + 1) * win <= (int64_t)gridDim.x) bid = grp * win + (within & 7) * CH + (within >> 3);
    }
    int64_t slice = bid * 4 + (threadIdx.x >> 6);
    if (slist) slice = slice < nlist ? (int64_t)slist[slice] : (int64_t)nslices;
    double y0 = 0, y1 = 0, yy2 = 0, z0 = 0, z1 = 0, z2 = 0;
    const int64_t row = slice < nslices ? (int64_t)rowof[slice * 64 + lane] : nloc;   // SELL-C-sigma: position -> block row
    if (slice < nslices) {
        const int32_t k0 = slot_ptr[slice], k1 = slot_ptr[slice + 1];
        const int32_t *cp = cols + (int64_t)k0 * 64 + lane;
        const VT *vp = vals + (int64_t)k0 * vstream<VT>::STRIDE + lane;
#define STAN_SPMV2_BLOCK(C, VP)                                                       \
    {                                                                                 \
        double a[9];                                                                  \
        load9<true, VT>(VP, a);                                                       \
        double x0 = x[3 * (C)], x1 = x[3 * (C) + 1], xx2 = x[3 * (C) + 2];            \
        double u0 = x2[3 * (C)], u1 = x2[3 * (C) + 1], u2 = x2[3 * (C) + 2];          \
        if (vstream<VT>::FX) {                                                        \
            x0 *= FX48_INV; x1 *= FX48_INV; xx2 *= FX48_INV;                          \
            u0 *= FX48_INV; u1 *= FX48_INV; u2 *= FX48_INV;                           \
        }                                                                             \
        y0 += a[0] * x0 + a[1] * x1 + a[2] * xx2;                                     \
        y1 += a[3] * x0 + a[4] * x1 + a[5] * xx2;                                     \
        yy2 += a[6] * x0 + a[7] * x1 + a[8] * xx2;                                    \
        z0 += a[0] * u0 + a[1] * u1 + a[2] * u2;                                      \
        z1 += a[3] * u0 + a[4] * u1 + a[5] * u2;                                      \
        z2 += a[6] * u0 + a[7] * u1 + a[8] * u2;                                      \
    }
        if (cs.packed && cs.ok[slice] == 1) {   // (two-base slices read the int32 columns here)
            const uint32_t *cq = cs.packed + (int64_t)cs.pair_ptr[slice] * 64 + lane;
            const int32_t *bp = cs.base + __builtin_amdgcn_readfirstlane(k0);
            int32_t k = k0;
            for (; k + 1 < k1; k += 2) {
                const uint32_t wd = ld_stream<true>(cq);
                const int64_t c = (int64_t)bp[0] + (int64_t)(wd & 0xffffu);
                const int64_t c2 = (int64_t)bp[1] + (int64_t)(wd >> 16);
                STAN_SPMV2_BLOCK(c, vp)
                STAN_SPMV2_BLOCK(c2, vp + vstream<VT>::STRIDE)
                cq += 64;
                bp += 2;
                vp += 2 * vstream<VT>::STRIDE;
            }
            if (k < k1) {
                const int64_t c = (int64_t)bp[0] + (int64_t)(ld_stream<true>(cq) & 0xffffu);
                STAN_SPMV2_BLOCK(c, vp)
            }
        } else {
#pragma unroll 2
            for (int32_t k = k0; k < k1; k++) {
                const int64_t c = ld_stream<true>(cp);
                STAN_SPMV2_BLOCK(c, vp)
                cp += 64;
                vp += vstream<VT>::STRIDE;
            }
        }
#undef STAN_SPMV2_BLOCK
        if (row < nloc) {
#if STAN_Y_NT  // both products are read exactly once, by k_step
            __builtin_nontemporal_store(y0, y + 3 * row); __builtin_nontemporal_store(y1, y + 3 * row + 1);
            __builtin_nontemporal_store(yy2, y + 3 * row + 2);
            __builtin_nontemporal_store(z0, y2 + 3 * row); __builtin_nontemporal_store(z1, y2 + 3 * row + 1);
            __builtin_nontemporal_store(z2, y2 + 3 * row + 2);
#else
            y[3 * row] = y0; y[3 * row + 1] = y1; y[3 * row + 2] = yy2;
            y2[3 * row] = z0; y2[3 * row + 1] = z1; y2[3 * row + 2] = z2;
#endif
        }
    }
    double d = 0;
    if (slice < nslices && row < nloc) d = y0 * x[3 * row] + y1 * x[3 * row + 1] + yy2 * x[3 * row + 2];
    const double t = block_sum(d, sh);
    if (threadIdx.x == 0) st_agent(partial + blockIdx.x + poff, t);
    if (fold.counter && fold_arrive(fold, &sh_last)) fold_finish<1>(fold, partial, sh);
}


// ---- folded rows (STAN_OPT_ROW_FOLDING, fold.hip) ---------------------------------------------------------
// The streams of fold.hip: padded-slot layout ([slot][ROWS][64]) with W ~ blocks/64 slots per slice; slots
// < own[lane] hold the lane's own row, the slots behind them a piece of ONE longer row of the slice (or zeros).
// Two accumulators per lane; the foreign ones go through LDS and each folded row adds its helpers' sums in a
// fixed order (descending lane).  NRHS = 1: k_spmv (DOT as there); NRHS = 2: k_spmv2.
template <typename VT, int DOT, int NRHS>
// at most 5 waves per SIMD (96 VGPRs): with the default target of 6 (80 VGPRs) the scheduler issues the gathers of a
// trip's second slot only after the first slot's data has arrived -- one more dependent round trip per trip
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, (NRHS == 2 ? 4 : 5))))
k_spmv_fold(int32_t nslices, int64_t nloc, const int32_t *__restrict__ fold_ptr, const int32_t *__restrict__ rowof,
            const uint32_t *__restrict__ meta, const int32_t *__restrict__ cols, const VT *__restrict__ vals,
            const double *__restrict__ x, const double *__restrict__ x2, double *__restrict__ y, double *__restrict__ y2,
            double *partial, const int64_t *st, int64_t kiter, const int32_t *__restrict__ slist, int32_t nlist,
            int32_t poff, fold_args fold, colstream cs) {
    __shared__ double sh[4];
    __shared__ int sh_last;
    __shared__ double fsh[4][3 * NRHS][64];
    if (stopped(st, kiter)) { fold_skip(fold); return; }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int64_t bid = blockIdx.x;
    {   // XCD-chunked workgroup mapping, as in k_spmv (variant 9)
        constexpr int CH = 32;
        const int64_t win = 8 * CH, grp = bid / win, within = bid - grp * win;
        if ((grp + 1) * win <= (int64_t)gridDim.x) bid = grp * win + (within & 7) * CH + (within >> 3);
    }
    int64_t slice = bid * 4 + w;
    if (slist) slice = slice < nlist ? (int64_t)slist[slice] : (int64_t)nslices;
    double y0 = 0, y1 = 0, yy2 = 0, z0 = 0, z1 = 0, z2 = 0;     // own row
    double f0 = 0, f1 = 0, f2 = 0, g0 = 0, g1 = 0, g2 = 0;      // the piece of a longer row this lane carries
    const int64_t row = slice < nslices ? (int64_t)rowof[slice * 64 + lane] : nloc;
    uint32_t m = 0;
    if (slice < nslices) {
        m = meta[slice * 64 + lane];
        const int32_t own = (int32_t)(m & 0xffffu);
        const int32_t k0 = fold_ptr[slice], k1 = fold_ptr[slice + 1];
        const int32_t *cp = cols + (int64_t)k0 * 64 + lane;
        const VT *vp = vals + (int64_t)k0 * vstream<VT>::STRIDE + lane;
// LOAD then MATH for both slots of a trip: with the two accumulator sets the scheduler otherwise
// issued the gathers of a trip's SECOND slot only after the first slot's data had arrived (one more dependent round
// trip per trip: the folded kernel was 3-6 % slower than k_spmv at equal slot counts).
#define STAN_FOLD_LOAD(S, C, VP)                                                                \
        double a##S[9];                                                                         \
        load9<true, VT>(VP, a##S);                                                              \
        double x0##S = x[3 * (C)], x1##S = x[3 * (C) + 1], x2##S = x[3 * (C) + 2];              \
        double u0##S = 0, u1##S = 0, u2##S = 0;                                                 \
        if (NRHS == 2) { u0##S = x2[3 * (C)]; u1##S = x2[3 * (C) + 1]; u2##S = x2[3 * (C) + 2]; }
#define STAN_FOLD_MATH(S, KL)                                                                   \
    {                                                                                           \
        if (vstream<VT>::FX) { x0##S *= FX48_INV; x1##S *= FX48_INV; x2##S *= FX48_INV; }       \
        const double t0 = a##S[0] * x0##S + a##S[1] * x1##S + a##S[2] * x2##S;                  \
        const double t1 = a##S[3] * x0##S + a##S[4] * x1##S + a##S[5] * x2##S;                  \
        const double t2 = a##S[6] * x0##S + a##S[7] * x1##S + a##S[8] * x2##S;                  \
        const bool mine = (KL) < own;                                                           \
        y0 += mine ? t0 : 0.0; y1 += mine ? t1 : 0.0; yy2 += mine ? t2 : 0.0;                   \
        f0 += mine ? 0.0 : t0; f1 += mine ? 0.0 : t1; f2 += mine ? 0.0 : t2;                    \
        if (NRHS == 2) {                                                                        \
            if (vstream<VT>::FX) { u0##S *= FX48_INV; u1##S *= FX48_INV; u2##S *= FX48_INV; }   \
            const double s0 = a##S[0] * u0##S + a##S[1] * u1##S + a##S[2] * u2##S;              \
            const double s1 = a##S[3] * u0##S + a##S[4] * u1##S + a##S[5] * u2##S;              \
            const double s2 = a##S[6] * u0##S + a##S[7] * u1##S + a##S[8] * u2##S;              \
            z0 += mine ? s0 : 0.0; z1 += mine ? s1 : 0.0; z2 += mine ? s2 : 0.0;                \
            g0 += mine ? 0.0 : s0; g1 += mine ? 0.0 : s1; g2 += mine ? 0.0 : s2;                \
        }                                                                                       \
    }
        if (cs.packed && cs.ok[slice] == 1) {   // wave-uniform: the folded copy has its own packed column stream (modes 0 / 1)
            const uint32_t *cq = cs.packed + (int64_t)cs.pair_ptr[slice] * 64 + lane;
            const int32_t *bp = cs.base + __builtin_amdgcn_readfirstlane(k0);
            int32_t k = k0;
            // the packed offsets of a trip are loaded one trip ahead: both gathers of a trip can go out with its values
            uint32_t wd = __builtin_nontemporal_load(cq);
            for (; k + 1 < k1; k += 2) {
                const uint32_t wn = __builtin_nontemporal_load(k + 2 < k1 ? cq + 64 : cq);
                const int64_t c = (int64_t)bp[0] + (int64_t)(wd & 0xffffu);
                const int64_t c2 = (int64_t)bp[1] + (int64_t)(wd >> 16);
                STAN_FOLD_LOAD(A, c, vp)
                STAN_FOLD_LOAD(B, c2, vp + vstream<VT>::STRIDE)
                STAN_FOLD_MATH(A, k - k0)
                STAN_FOLD_MATH(B, k - k0 + 1)
                cq += 64;
                bp += 2;
                vp += 2 * vstream<VT>::STRIDE;
                wd = wn;
            }
            if (k < k1) {
                const int64_t c = (int64_t)bp[0] + (int64_t)(wd & 0xffffu);
                STAN_FOLD_LOAD(A, c, vp)
                STAN_FOLD_MATH(A, k - k0)
            }
        } else {
            int32_t k = k0;
            for (; k + 1 < k1; k += 2) {
                const int64_t c = __builtin_nontemporal_load(cp), c2 = __builtin_nontemporal_load(cp + 64);
                STAN_FOLD_LOAD(A, c, vp)
                STAN_FOLD_LOAD(B, c2, vp + vstream<VT>::STRIDE)
                STAN_FOLD_MATH(A, k - k0)
                STAN_FOLD_MATH(B, k - k0 + 1)
                cp += 128;
                vp += 2 * vstream<VT>::STRIDE;
            }
            if (k < k1) {
                const int64_t c = __builtin_nontemporal_load(cp);
                STAN_FOLD_LOAD(A, c, vp)
                STAN_FOLD_MATH(A, k - k0)
            }
        }
#undef STAN_FOLD_LOAD
#undef STAN_FOLD_MATH
    }
    // The exchange is wave-local (a slice is one wavefront): the LDS executes one wave's instructions in order, so
    // the reads below see the writes above without a workgroup barrier -- the four slices of a workgroup do not
    // wait for each other here; a slice without folded rows skips it altogether.
    const int nh = (int)(m >> 24);
    if (__any(nh > 0)) {
        fsh[w][0][lane] = f0; fsh[w][1][lane] = f1; fsh[w][2][lane] = f2;
        if (NRHS == 2) { fsh[w][3][lane] = g0; fsh[w][4][lane] = g1; fsh[w][5][lane] = g2; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int hfirst = (int)((m >> 16) & 0xffu);
        for (int h = 0; h < nh; h++) {   // a folded row: own part + its pieces, last lane first
            const int j = hfirst - h;
            y0 += fsh[w][0][j]; y1 += fsh[w][1][j]; yy2 += fsh[w][2][j];
            if (NRHS == 2) { z0 += fsh[w][3][j]; z1 += fsh[w][4][j]; z2 += fsh[w][5][j]; }
        }
    }
    if (slice < nslices) {
        if (row < nloc) {
#if STAN_Y_NT
            __builtin_nontemporal_store(y0, y + 3 * row); __builtin_nontemporal_store(y1, y + 3 * row + 1);
            __builtin_nontemporal_store(yy2, y + 3 * row + 2);
            if (NRHS == 2) {
                __builtin_nontemporal_store(z0, y2 + 3 * row); __builtin_nontemporal_store(z1, y2 + 3 * row + 1);
                __builtin_nontemporal_store(z2, y2 + 3 * row + 2);
            }
#else
            y[3 * row] = y0; y[3 * row + 1] = y1; y[3 * row + 2] = yy2;
            if (NRHS == 2) { y2[3 * row] = z0; y2[3 * row + 1] = z1; y2[3 * row + 2] = z2; }
#endif
        }
    }
    if (DOT) {
        double d = 0, e = 0;
        if (slice < nslices && row < nloc) {
            const double x0 = x[3 * row], x1 = x[3 * row + 1], xx2 = x[3 * row + 2];
            d = y0 * x0 + y1 * x1 + yy2 * xx2;
            if (DOT == 2) e = x0 * x0 + x1 * x1 + xx2 * xx2;
        }
        const double t = block_sum(d, sh);
        if (DOT == 2) {
            const double u = block_sum(e, sh);
            if (threadIdx.x == 0) {
                st_agent(partial + 2 * (int64_t)(blockIdx.x + poff), u);
                st_agent(partial + 2 * (int64_t)(blockIdx.x + poff) + 1, t);
            }
        } else if (threadIdx.x == 0) st_agent(partial + blockIdx.x + poff, t);
        if (fold.counter && fold_arrive(fold, &sh_last)) fold_finish<(DOT == 2 ? 2 : 1)>(fold, partial, sh);
    }
}

// ---- CG step kernels ----------------------------------------------------------------------------

struct step_args {
    int64_t n3;           // 3 * owned block rows
    int64_t k;            // iteration number (1-based)
    double *sc;           // scalars
    int64_t *st;          // status
    const double *xcur;   // rx
    double *xnext;        // cx
    double *r;            // r (in), cr (out)
    const double *p;
    const double *v;      // A^ p
    const double *bh;
    double *partial;      // [blocks][2]: r2, merit
    const double *w;      // A^ x (fused refresh)
    int merit;            // 0: the merit-function stop is off, skip its sum (and the b^ read)
    int refresh;          // 0: r -= a v; 1: only cx is formed here (r from a second SpMV);
                          // 2: fused refresh, r = b^ - (w + a v) with w = A^ x from the same pass
    int defer_x;          // 1: x' = x + a p is formed by k_update (one read of p for both updates); only
                          //    when the merit sum is off and the iteration needs no x' before k_update
    fold_args fold;       // r.r and the merit sum are added up by the last block (-> sc[S_R2NEW..])
    red_src rs_vmv;       // where p.Ap is found (p2p_device.h)
};

template <bool RNT>
__global__ void __launch_bounds__(VEC_T) k_step(step_args a) {
    __shared__ double sh[4];
    __shared__ int sh_last;
    if (stopped(a.st, a.k)) { if (a.refresh != 1) fold_skip(a.fold); return; }
    double vmv_[1];
    red_get<1>(a.sc + S_VMV, a.rs_vmv, vmv_, sh);
    const double vmv = vmv_[0];
    const double rho = a.sc[S_RHO0 + (a.k & 1)];
    int bad = 0;
    if (!isfinite(vmv) || vmv <= 0) bad = isfinite(vmv) ? -5 : -4;
    const double alpha = rho / vmv;
    if (!bad && !isfinite(alpha)) bad = -4;
    if (bad) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            a.st[T_TYPE] = bad;
            a.st[T_ITERS] = a.k;
            a.st[T_XSEL] = (a.k - 1) & 1;  // rx of the previous iteration
            a.st[T_ITER_B] = a.k;
        }
        if (a.refresh != 1) fold_skip(a.fold);
        return;
    }
    double s_r2 = 0, s_mf = 0;
    const int64_t stride = (int64_t)gridDim.x * VEC_T;
    for (int64_t i = (int64_t)blockIdx.x * VEC_T + threadIdx.x; i < a.n3; i += stride) {
        double cx = 0;
        if (!a.defer_x) {
#if STAN_VEC_NT  // the iterate vectors pass through once per kernel: keep them out of the caches p lives in
            const double pi = __builtin_nontemporal_load(a.p + i);
            cx = __builtin_nontemporal_load(a.xcur + i) + alpha * pi;
            __builtin_nontemporal_store(cx, a.xnext + i);
#else
            const double pi = a.p[i];
            cx = a.xcur[i] + alpha * pi;
            a.xnext[i] = cx;
#endif
        }
        if (a.refresh == 0) {
#if STAN_VEC_NT
            const double cr = __builtin_nontemporal_load(a.r + i) - alpha * __builtin_nontemporal_load(a.v + i);
#else
            const double cr = a.r[i] - alpha * a.v[i];
#endif
            if (RNT) __builtin_nontemporal_store(cr, a.r + i);
            else a.r[i] = cr;
            s_r2 += cr * cr;
            if (a.merit) s_mf -= (cr + a.bh[i]) * cx;
        } else if (a.refresh == 2) {
            const double b = a.bh[i];
            const double mv = a.w[i] + alpha * a.v[i];  // A^ (x + a p)
            const double cr = b - mv;
            if (RNT) __builtin_nontemporal_store(cr, a.r + i);
            else a.r[i] = cr;
            s_r2 += cr * cr;
            if (a.merit) s_mf += (mv - 2 * b) * cx;
        }
    }
    if (a.refresh != 1) {
        const double t0 = block_sum(s_r2, sh);
        const double t1 = block_sum(s_mf, sh);
        if (threadIdx.x == 0) {
            st_agent(a.partial + 2 * blockIdx.x, t0);
            st_agent(a.partial + 2 * blockIdx.x + 1, t1);
        }
        if (a.fold.counter && fold_arrive(a.fold, &sh_last)) fold_finish<2>(a.fold, a.partial, sh);
    }
}

// refresh iterations: r = b^ - A^ cx, merit = sum (mv - 2 b^) cx
__global__ void __launch_bounds__(VEC_T)
k_refresh(int64_t n3, int64_t k, const int64_t *st, const double *bh, const double *mv,
          const double *cx, double *r, double *partial, fold_args fold) {
    __shared__ double sh[4];
    __shared__ int sh_last;
    if (st[T_ITER_A] < k || st[T_ITER_B] <= k) { fold_skip(fold); return; }
    double s_r2 = 0, s_mf = 0;
    const int64_t stride = (int64_t)gridDim.x * VEC_T;
    for (int64_t i = (int64_t)blockIdx.x * VEC_T + threadIdx.x; i < n3; i += stride) {
        const double b = bh[i], m = mv[i];
        const double cr = b - m;
#if STAN_R_NT
        __builtin_nontemporal_store(cr, r + i);
#else
        r[i] = cr;
#endif
        s_r2 += cr * cr;
        s_mf += (m - 2 * b) * cx[i];
    }
    const double t0 = block_sum(s_r2, sh);
    const double t1 = block_sum(s_mf, sh);
    if (threadIdx.x == 0) {
        st_agent(partial + 2 * blockIdx.x, t0);
        st_agent(partial + 2 * blockIdx.x + 1, t1);
    }
    if (fold.counter && fold_arrive(fold, &sh_last)) fold_finish<2>(fold, partial, sh);
}

// decisions of the iteration + p = r + beta p
template <bool PNT>
__global__ void __launch_bounds__(VEC_T)
k_update(int64_t n3, int64_t k, double *sc, int64_t *st, double epsf, int64_t maxits,
         int64_t its_before_restart, int merit_stop, const double *r, double *p,
         const double *xcur, double *xnext /* non-null: the deferred x' = x + a p of this iteration */,
         red_src rs_r2, red_src rs_vmv) {
    __shared__ double sh[4];
    if (st[T_ITER_A] < k || st[T_ITER_B] <= k) return;
    double t2[2];
    red_get<2>(sc + S_R2NEW, rs_r2, t2, sh);   // r.r, merit sum
    const double r2 = t2[0], merit = t2[1];
    const double rho = sc[S_RHO0 + (k & 1)], prevmf = sc[S_PMF0 + (k & 1)];
    const double bnorm = sc[S_BNORM];
    int type = 0;
    int64_t xsel = k & 1;  // cx lives in buffer k&1
    if (sqrt(r2) <= epsf * bnorm) type = 1;
    else if (k >= maxits && maxits > 0) type = 5;
    else if (merit_stop && merit >= prevmf) { type = 7; xsel = (k - 1) & 1; }
    double beta = 0;
    const bool restart = (k % its_before_restart) == 0;
    if (!type && !restart) {
        beta = r2 / rho;
        if (!isfinite(beta)) { type = -4; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc[S_R2OUT] = r2;
        if (type) {
            st[T_TYPE] = type;
            st[T_ITERS] = k;
            st[T_XSEL] = xsel;
            st[T_ITER_A] = k;
        } else {
            sc[S_RHO0 + ((k + 1) & 1)] = r2;
            sc[S_PMF0 + ((k + 1) & 1)] = merit;
        }
    }
    const int64_t stride = (int64_t)gridDim.x * VEC_T;
    // alpha of this iteration, as k_step formed it (same operands, same bits)
    double alpha = 0.0;
    if (xnext) {   // block-uniform
        double v1[1];
        red_get<1>(sc + S_VMV, rs_vmv, v1, sh);
        alpha = rho / v1[0];
    }
    if (type) {
        // the iteration that stops (types 1, 5, -4 select x' = buffer k & 1) still owes its x'
        if (xnext && xsel == (k & 1))
            for (int64_t i = (int64_t)blockIdx.x * VEC_T + threadIdx.x; i < n3; i += stride)
                __builtin_nontemporal_store(__builtin_nontemporal_load(xcur + i) + alpha * __builtin_nontemporal_load(p + i), xnext + i);
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * VEC_T + threadIdx.x; i < n3; i += stride)
    {
        // PNT: p is read, rewritten and not touched again until the next product gathers it: take it
        // past the caches both ways (a non-temporal store to a line a plain load has just brought
        // into L2 only dirties that line: lab modes 9-12 of tools/fold_ab.py)
        const double po = PNT ? __builtin_nontemporal_load(p + i) : p[i];
        const double pn = (PNT ? __builtin_nontemporal_load(r + i) : r[i]) + beta * po;
        if (PNT) __builtin_nontemporal_store(pn, p + i);
        else p[i] = pn;
        if (xnext) __builtin_nontemporal_store(__builtin_nontemporal_load(xcur + i) + alpha * po, xnext + i);
    }
}

// ---- single-reduction CG (Chronopoulos-Gear), STAN_OPT_CG_SINGLE_REDUCE --------------------------
// Same iterates as the classic loop in exact arithmetic, ONE reduction point per iteration:
//   w_k = A r_k,  gamma_k = r_k.r_k,  delta_k = r_k.w_k          (the SpMV, both sums in its epilogue)
//   beta_k = gamma_k / gamma_{k-1},   p_k.A p_k = delta_k - beta_k gamma_k / alpha_{k-1},
//   alpha_k = gamma_k / p_k.A p_k
//   p = r + beta p,  s = w + beta s (= A p),  x' = x + alpha p,  r' = r - alpha s
// Per iteration: this kernel + one SpMV (+ ONE all-reduce of 3 doubles when sharded) instead of
// SpMV, all-reduce, step, all-reduce, update.  Iteration k's kernel first takes the decisions of
// iteration k-1 (its residual norm and merit value arrived with the last reduction), with the
// codes and the "previous point on type 7" rule of the classic loop.  Rounding differs from the
// classic recurrences (alpha comes from a three-term formula), so iteration counts may differ by
// a few: the default stays the classic loop, which is the oracle's.
struct sr_args {
    int64_t n3, k;
    double *sc;
    int64_t *st;
    double epsf;
    int64_t maxits, its_before_restart;
    int merit_stop;
    int refresh;          // 1: r' comes from b^ - A x' (k_refresh), only p, s, x' are formed here
    const double *xcur;
    double *xnext;
    double *r, *p, *s;
    const double *w, *bh;
    double *partial;      // [blocks] merit partials
    fold_args fold;       // -> sc[S_SR_MERIT]
    red_src rs;           // where gamma, delta, merit of the last reduction are found
};

__global__ void __launch_bounds__(VEC_T) k_vec_sr(sr_args a) {
    __shared__ double sh[4];
    __shared__ int sh_last;
    const int64_t k = a.k;
    if (a.st[T_ITER_A] < k) return;   // stopped by an earlier iteration's decisions
    double t3[3];
    red_get<3>(a.sc + S_SR_GAMMA, a.rs, t3, sh);
    const double gamma = t3[0], delta = t3[1], merit = t3[2];
    const double bnorm = a.sc[S_BNORM];
    int type = 0;
    int64_t its = k - 1, xsel = (k - 1) & 1;
    if (!isfinite(gamma)) type = -4;
    else if (k > 1) {   // decisions of iteration k-1 (x_{k-1} lives in buffer (k-1)&1)
        if (sqrt(gamma) <= a.epsf * bnorm) type = 1;
        else if (k - 1 >= a.maxits && a.maxits > 0) type = 5;
        else if (a.merit_stop && merit >= a.sc[S_PMF0 + ((k - 1) & 1)]) { type = 7; xsel = (k - 2) & 1; }
    }
    double alpha = 0, beta = 0;
    if (!type) {
        const bool restart = k == 1 || ((k - 1) % a.its_before_restart) == 0;
        double pap = delta;
        if (!restart) {
            beta = gamma / a.sc[S_SR_GP0 + ((k - 1) & 1)];
            pap = delta - beta * gamma / a.sc[S_SR_AP0 + ((k - 1) & 1)];
        }
        if (!isfinite(pap) || !isfinite(beta)) type = -4;
        else if (pap <= 0) type = -5;
        else { alpha = gamma / pap; if (!isfinite(alpha)) type = -4; }
        if (type) { its = k; xsel = (k - 1) & 1; }   // as k_step: the previous point, this iteration's number
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.sc[S_R2OUT] = gamma;
        if (type) {
            a.st[T_TYPE] = type;
            a.st[T_ITERS] = its;
            a.st[T_XSEL] = xsel;
            a.st[T_ITER_A] = k - 1;   // this iteration's product and everything later return at once
        } else {
            a.sc[S_SR_GP0 + (k & 1)] = gamma;
            a.sc[S_SR_AP0 + (k & 1)] = alpha;
            a.sc[S_PMF0 + (k & 1)] = merit;
        }
    }
    if (type) return;
    double s_mf = 0;
    const int64_t stride = (int64_t)gridDim.x * VEC_T;
    for (int64_t i = (int64_t)blockIdx.x * VEC_T + threadIdx.x; i < a.n3; i += stride) {
        const double ri = a.r[i];
        const double pi = ri + beta * a.p[i];
        const double si = __builtin_nontemporal_load(a.w + i) + beta * a.s[i];
        const double cx = __builtin_nontemporal_load(a.xcur + i) + alpha * pi;
        __builtin_nontemporal_store(pi, a.p + i);
        __builtin_nontemporal_store(si, a.s + i);
        __builtin_nontemporal_store(cx, a.xnext + i);
        if (!a.refresh) {
            const double cr = ri - alpha * si;
            __builtin_nontemporal_store(cr, a.r + i);   // gathered by the product that follows (see STAN_P_NT)
            if (a.merit_stop) s_mf -= (cr + a.bh[i]) * cx;
        }
    }
    if (!a.refresh) {
        const double t = block_sum(s_mf, sh);
        if (threadIdx.x == 0) st_agent(a.partial + blockIdx.x, t);
        if (a.fold.counter && fold_arrive(a.fold, &sh_last)) fold_finish<1>(a.fold, a.partial, sh);
    }
}

// U[d - red[d]] = s_d * x^_d on free DOFs (SolverFunctions.cs:305 lincgresults + un-scaling)
__global__ void k_result(int64_t n3, int64_t dof0, const int32_t *red, const double *s,
                         const double *x, double *U) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += stride) {
        const int32_t rd = red[dof0 + i];
        if (rd != -1) U[dof0 + i - rd] = s[i] * x[i];
    }
}
// multi-rank: scaled solution into the global block vector, compressed after the all-gather
__global__ void k_result_full(int64_t n3, const double *s, const double *x, double *full) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += stride)
        full[i] = s[i] * x[i];
}
__global__ void k_compress(int64_t n_dof, const int32_t *red, const double *full, double *U) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_dof; i += stride) {
        const int32_t rd = red[i];
        if (rd != -1) U[i - rd] = full[i];
    }
}
// reduced host-order vector <-> full local vector (for stan_hip_spmv)
__global__ void k_expand(int64_t n3, int64_t dof0, const int32_t *red, const double *in,
                         const double *div, double *out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += stride) {
        const int32_t rd = red[dof0 + i];
        double v = rd == -1 ? 0.0 : in[dof0 + i - rd];
        if (div) v /= div[i];
        out[i] = v;
    }
}
__global__ void k_compress_div(int64_t n3, const int32_t *red, const double *s, const double *y,
                               double *out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += stride) {
        const int32_t rd = red[i];
        if (rd != -1) out[i - rd] = s ? y[i] / s[i] : y[i];
    }
}
__global__ void k_fill(double *p, int64_t n, double v) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }
inline unsigned vec_grid(int64_t n) {
    int64_t b = (n + VEC_T - 1) / VEC_T;
    if (b < 1) b = 1;
    return (unsigned)(b > VEC_BLOCKS ? VEC_BLOCKS : b);
}

struct dev_bufs {
    stan_ctx *ctx = nullptr;
    std::vector<void *> p;
    ~dev_bufs() {
        for (void *q : p) stan_dfree(ctx, q);
    }
};
template <typename T>
int alloc(stan_ctx *ctx, dev_bufs &b, T **p, size_t n) {
    b.ctx = ctx;
    int rc = stan_dmalloc(ctx, p, n);
    if (rc == STAN_OK) b.p.push_back((void *)*p);
    return rc;
}

// k_spmv_small instead of k_spmv: decided by the GLOBAL number of block rows, so that a shard and the
// whole matrix sum their rows in the same order
inline bool stan_small_system(const stan_ctx *ctx, const stan_matrix *K) {
    return ctx->spmv_variant < 0 && K->nb_glob <= ctx->spmv_small_rows;
}

// the folded form of the value stream `vals` of K, if the products are to read it (fold.hip)
template <typename VT> const VT *fold_vals(const stan_ctx *ctx, const stan_matrix *K, const VT *vals);
template <> const double *fold_vals<double>(const stan_ctx *ctx, const stan_matrix *K, const double *vals) {
    if (ctx->fold_probe && (const void *)vals == ctx->fold_probe) return vals;
    return ctx->row_folding != 0 && vals == K->d_vals ? K->d_fold_vals : nullptr;
}
template <> const float *fold_vals<float>(const stan_ctx *ctx, const stan_matrix *K, const float *vals) {
    if (ctx->fold_probe && (const void *)vals == ctx->fold_probe) return vals;
    return ctx->row_folding != 0 && vals == K->d_vals32 ? K->d_fold_vals32 : nullptr;
}
template <> const uint32_t *fold_vals<uint32_t>(const stan_ctx *ctx, const stan_matrix *K, const uint32_t *vals) {
    if (ctx->fold_probe && (const void *)vals == ctx->fold_probe) return vals;
    return ctx->row_folding != 0 && vals == K->d_vals48 ? K->d_fold_vals48 : nullptr;
}
inline colstream fold_cols_of(const stan_ctx *ctx, const stan_matrix *K) {
    return ctx->cols16 && K->d_fold_cols16 ? make_colstream(K->d_fold_cols16, K->d_fold_colbase, K->d_fold_pair_ptr, K->d_fold_packed, K->nfslots)
                                           : NO_COLSTREAM;
}
// which: 0 = all slices, 1 = interior list, 2 = boundary list (partials offset by the
// interior launch's block count).  Returns the number of partial slots this launch writes.
// `fold`: counter/out of the folded reduction (counter == nullptr: partials only); nblocks and np
// are filled in here (np = this launch's slots + poff: the boundary launch of a split product
// also adds up what the interior launch left).
template <typename VT, int DOT>
unsigned launch_spmv(stan_ctx *ctx, stan_matrix *K, const VT *vals, const double *x, double *y,
                     double *partial, const int64_t *st, int64_t k, int which = 0,
                     hipStream_t stream = nullptr, fold_args fold = NO_FOLD) {
    if (!stream) stream = ctx->stream;
    const int32_t *slist = which == 1 ? K->d_sl_int : which == 2 ? K->d_sl_bnd : nullptr;
    const int32_t nlist = which == 1 ? K->n_sl_int : which == 2 ? K->n_sl_bnd : K->nslices;
    const colstream cs = ctx->cols16 && K->d_cols16 ? make_colstream(K->d_cols16, K->d_colbase, K->d_pair_ptr, K->d_slice_packed, K->nslots)
                                                    : NO_COLSTREAM;
    if (stan_small_system(ctx, K)) {   // one workgroup per slice (k_spmv_small); partials per slice
        const int32_t poff_s = which == 2 ? K->n_sl_int : 0;
        const unsigned grid_s = (unsigned)nlist;
        if (grid_s == 0) return 0;
        fold.nblocks = grid_s;
        fold.np = (int)grid_s + poff_s;
        hipLaunchKernelGGL((k_spmv_small<VT, DOT>), dim3(grid_s), dim3(256), 0, stream, K->nslices, K->nloc,
                           K->d_slot_ptr, K->d_rowof, K->d_cols, vals, x, y, partial, st, k, slist, nlist, poff_s, fold, cs);
        return grid_s;
    }
    const int32_t poff = which == 2 ? (int32_t)nblk(K->n_sl_int, 4) : 0;
    const unsigned grid = nblk(nlist, 4);
    if (grid == 0) return 0;
    fold.nblocks = grid;
    fold.np = (int)grid + poff;
    if (const VT *fv = fold_vals<VT>(ctx, K, vals)) {
        hipLaunchKernelGGL((k_spmv_fold<VT, DOT, 1>), dim3(grid), dim3(256), 0, stream, K->nslices, K->nloc, K->d_fold_ptr,
                           K->d_rowof, K->d_fold_meta, K->d_fold_cols, fv, x, (const double *)nullptr, y, (double *)nullptr,
                           partial, st, k, slist, nlist, poff, fold, fold_cols_of(ctx, K));
        return grid;
    }
#define SPMV_CASE(V)                                                                          \
    case V:                                                                                   \
        hipLaunchKernelGGL((k_spmv<VT, DOT, V>), dim3(grid), dim3(256), 0, stream, K->nslices, \
                           K->nloc, K->d_slot_ptr, K->d_rowof, K->d_cols, vals, x, y, partial, st, k, \
                           slist, nlist, poff, fold, cs);                                     \
        break;
    // auto (-1): non-temporal matrix stream + XCD-chunked workgroup mapping (variant 9), with the
    // loop unrolled by 4 for the FIXED-48 stream, whose iterations carry 21 % fewer bytes in
    // flight (variant 12).  One process, one box (tools/fx48_variants.py, min of 3 x 20 launches):
    // fp64 plain 1.181 / nt 1.102 / nt+chunk 1.100 ms; FIXED-48 1.003 / 0.955 / nt+unroll4 0.925;
    // fp32 0.636 / 0.583 / 0.580.
    const int variant = ctx->spmv_variant >= 0 ? ctx->spmv_variant : vstream<VT>::FX ? 12 : 9;
    switch (variant) {
        SPMV_CASE(9) SPMV_CASE(12)
#ifdef STAN_LAB
        SPMV_CASE(1) SPMV_CASE(2) SPMV_CASE(3) SPMV_CASE(4) SPMV_CASE(5) SPMV_CASE(6) SPMV_CASE(7)
        SPMV_CASE(8) SPMV_CASE(10) SPMV_CASE(11) SPMV_CASE(13) SPMV_CASE(14) SPMV_CASE(15) SPMV_CASE(16)
        SPMV_CASE(17) SPMV_CASE(18)
#endif
        default:
        SPMV_CASE(0)
    }
#undef SPMV_CASE
    return grid;
}

template <typename VT>
unsigned launch_spmv2(stan_ctx *ctx, stan_matrix *K, const VT *vals, const double *x, const double *x2,
                      double *y, double *y2, double *partial, const int64_t *st, int64_t k, int which,
                      hipStream_t stream, fold_args fold) {
    const int32_t *slist = which == 1 ? K->d_sl_int : which == 2 ? K->d_sl_bnd : nullptr;
    const int32_t nlist = which == 1 ? K->n_sl_int : which == 2 ? K->n_sl_bnd : K->nslices;
    const int32_t poff = which == 2 ? (int32_t)nblk(K->n_sl_int, 4) : 0;
    const unsigned grid = nblk(nlist, 4);
    if (grid == 0) return 0;
    fold.nblocks = grid;
    fold.np = (int)grid + poff;
    if (const VT *fv = fold_vals<VT>(ctx, K, vals)) {
        hipLaunchKernelGGL((k_spmv_fold<VT, 1, 2>), dim3(grid), dim3(256), 0, stream, K->nslices, K->nloc, K->d_fold_ptr,
                           K->d_rowof, K->d_fold_meta, K->d_fold_cols, fv, x, x2, y, y2, partial, st, k, slist, nlist, poff, fold,
                           fold_cols_of(ctx, K));
        return grid;
    }
    const colstream cs = ctx->cols16 && K->d_cols16 ? make_colstream(K->d_cols16, K->d_colbase, K->d_pair_ptr, K->d_slice_packed, K->nslots)
                                                    : NO_COLSTREAM;
    hipLaunchKernelGGL((k_spmv2<VT>), dim3(grid), dim3(256), 0, stream, K->nslices, K->nloc, K->d_slot_ptr,
                       K->d_rowof, K->d_cols, vals, x, x2, y, y2, partial, st, k, slist, nlist, poff, fold, cs);
    return grid;
}

// value-stream dispatch of one product (DOT as in k_spmv)
template <int DOT>
unsigned launch_spmv_any(stan_ctx *ctx, stan_matrix *K, int stream_kind, const double *x, double *y,
                         double *partial, const int64_t *st, int64_t k, int which, hipStream_t s,
                         fold_args fold) {
    if (stream_kind == STAN_PREC_MIXED)
        return launch_spmv<float, DOT>(ctx, K, K->d_vals32, x, y, partial, st, k, which, s, fold);
    if (stream_kind == STAN_PREC_FIXED48)
        return launch_spmv<uint32_t, DOT>(ctx, K, K->d_vals48, x, y, partial, st, k, which, s, fold);
    return launch_spmv<double, DOT>(ctx, K, K->d_vals, x, y, partial, st, k, which, s, fold);
}

}  // namespace

// Vectors of the CG for a matrix of K's sizes, owned by the context (see stan_cg_ws).
int stan_cg_workspace(stan_ctx *ctx, const stan_matrix *K) {
    const int64_t npad = (int64_t)K->nslices * 64;
    const int64_t ng = 3 * ((npad > K->nloc + K->nhalo ? npad : K->nloc + K->nhalo));
    const int64_t n3 = 3 * K->nloc > 0 ? 3 * K->nloc : 1;
    stan_cg_ws &ws = ctx->ws;
    if (ws.p && ws.ng >= ng && ws.n3 >= n3 && ws.ng <= ng + ng / 2 + 64) return STAN_OK;
    stan_cg_workspace_free(ctx);
    for (double **q : {&ws.xb[0], &ws.xb[1], &ws.p, &ws.r}) STANCHK(stan_dmalloc(ctx, q, (size_t)(ng > 0 ? ng : 1)));
    for (double **q : {&ws.v, &ws.w, &ws.bh, &ws.sv}) STANCHK(stan_dmalloc(ctx, q, (size_t)n3));
    ws.ng = ng; ws.n3 = n3;
    return STAN_OK;
}
void stan_cg_workspace_free(stan_ctx *ctx) {
    stan_cg_ws &ws = ctx->ws;
    for (double **q : {&ws.xb[0], &ws.xb[1], &ws.p, &ws.r, &ws.v, &ws.w, &ws.bh, &ws.sv}) {
        if (*q) stan_dfree(ctx, *q);
        *q = nullptr;
    }
    ws.ng = ws.n3 = 0;
}

// Last resort of the placement search (placement.hip): no value candidate was clear of the vectors'
// group, and all of them are still allocated -- so NEW vectors, allocated now, lie beyond them.
// They are taken straight from the driver (a parked pool block would be the old memory again); the
// caller probes with them and keeps them (commit) or not.
int stan_cg_workspace_move(stan_ctx *ctx, const stan_matrix *K, bool commit, stan_cg_ws *saved) {
    stan_cg_ws &ws = ctx->ws;
    if (saved && !commit && saved->p == nullptr) {          // step 1: swap fresh vectors in, keep the old ones in *saved
        if (!ws.p) return STAN_OK;
        *saved = ws;
        stan_cg_ws nw;
        nw.ng = ws.ng; nw.n3 = ws.n3;
        bool ok = true;
        double **dst[8] = {&nw.xb[0], &nw.xb[1], &nw.p, &nw.r, &nw.v, &nw.w, &nw.bh, &nw.sv};
        for (int i = 0; i < 8 && ok; i++)
            ok = hipMalloc((void **)dst[i], (size_t)(i < 4 ? (nw.ng > 0 ? nw.ng : 1) : nw.n3) * 8) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            for (double **q : dst) if (*q) hipFree(*q);
            saved->p = nullptr;
            return STAN_OK;   // no memory for it: keep what we have
        }
        ws = nw;
        (void)K;
        return STAN_OK;
    }
    if (!saved || saved->p == nullptr) return STAN_OK;
    // step 2: keep the new vectors (release the old ones for real) or go back to the old ones
    stan_cg_ws &drop = commit ? *saved : ws;
    stan_cg_ws keep = commit ? ws : *saved;
    for (double *q : {drop.xb[0], drop.xb[1], drop.p, drop.r, drop.v, drop.w, drop.bh, drop.sv}) {
        if (!q) continue;
        ctx->pool.live.erase((void *)q);
        hipFree(q);
    }
    ws = keep;
    if (commit && ctx->pool.enabled)   // the new blocks are pooled-class blocks from now on
        for (double *q : {ws.xb[0], ws.xb[1], ws.p, ws.r}) if ((size_t)ws.ng * 8 >= stan_pool::MIN_BYTES) ctx->pool.live[(void *)q] = (size_t)ws.ng * 8;
    if (commit && ctx->pool.enabled)
        for (double *q : {ws.v, ws.w, ws.bh, ws.sv}) if ((size_t)ws.n3 * 8 >= stan_pool::MIN_BYTES) ctx->pool.live[(void *)q] = (size_t)ws.n3 * 8;
    saved->p = nullptr;
    return STAN_OK;
}

// Diagonal scaling of the matrix (once per matrix): A^ = S K S.
static int ensure_scaled(stan_ctx *ctx, stan_matrix *K) {
    if (K->scaled) return STAN_OK;
    const int64_t npad = (int64_t)K->nslices * 64;
    const int64_t ns = 3 * (npad + K->nhalo);
    if (!K->d_scale) STANCHK(stan_dmalloc(ctx, &K->d_scale, (size_t)ns));
    hipLaunchKernelGGL(k_fill, dim3(vec_grid(ns)), dim3(VEC_T), 0, ctx->stream, K->d_scale, ns, 1.0);
    if (K->nloc > 0)
        hipLaunchKernelGGL(k_diag_scale, dim3(nblk(K->nloc, 256)), dim3(256), 0, ctx->stream, K->nloc,
                           K->d_rowlen, K->d_posof, K->d_slot_ptr, K->d_cols, K->d_vals, K->d_scale);
    if (ctx->comm || ctx->nranks > 1) {
        if (ctx->comm_p2p && ctx->p2p) {
            // Peer to peer, my neighbours write their rows straight into my halo region, which the k_fill above
            // has just initialised: nobody may write before everybody has done that.  One empty reduction
            // (every rank counts itself into every rank's counter, every stream waits for all) orders it; the
            // exchanges inside the CG are ordered by the loop's own reductions.  (Found with the ranks in
            // separate processes, where mapping the peers' vectors delays some ranks by milliseconds: a
            // neighbour's scaling factors arrived before the fill and were overwritten with 1.0.)
            const p2p_out po{stan_p2p_table(ctx), stan_p2p_reduce_slot(ctx), 3, 1};
            hipLaunchKernelGGL(k_reduce<1>, dim3(1), dim3(256), 0, ctx->stream, (const double *)nullptr, 0, (double *)nullptr, po, (const int64_t *)nullptr, (int64_t)0);
            STANCHK(stan_p2p_reduce_wait(ctx));
        }
        STANCHK(stan_comm_halo_exchange(ctx, K, K->d_scale));
    }
    if (K->nslices > 0)
        hipLaunchKernelGGL(k_scale_matrix, dim3(nblk(K->nslices, 4)), dim3(256), 0, ctx->stream,
                           K->nslices, K->d_slot_ptr, K->d_rowof, K->d_cols, K->d_vals, K->d_scale, 0);
    HIPCHK(ctx, hipGetLastError());
    K->scaled = true;
    // copies of the value stream made before the scaling (stan_hip_spmv_bench on a fresh matrix)
    // hold the unscaled K: a later solve must not iterate on them
    if (K->d_vals32) { stan_dfree(ctx, K->d_vals32); K->d_vals32 = nullptr; }
    if (K->d_vals48) { stan_dfree(ctx, K->d_vals48); K->d_vals48 = nullptr; }
    stan_matrix_drop_folded_values(ctx, K);
    K->fx48_refused = false;
    return STAN_OK;
}

int stan_matrix_make_fp32(stan_ctx *ctx, stan_matrix *K) {
    if (K->d_vals32) return STAN_OK;
    const int64_t n = K->nslots * 9 * 64;
    STANCHK(stan_dmalloc_streamed(ctx, (void **)&K->d_vals32, (size_t)n * 4,
                                  [&](const void *q, float *ms, bool self) {
                                      return stan_spmv_probe(ctx, K, q, (size_t)n * 4, STAN_PREC_MIXED, ms, self);
                                  }));
    hipLaunchKernelGGL(k_to_fp32, dim3(vec_grid(n) * 4), dim3(VEC_T), 0, ctx->stream, K->d_vals,
                       K->d_vals32, n);
    HIPCHK(ctx, hipGetLastError());
    return STAN_OK;
}

// FIXED-48 copy of the scaled values.  Returns STAN_OK with K->d_vals48 == nullptr when some
// entry is not representable (K not SPD): the caller then streams the fp64 values.
int stan_matrix_make_fx48(stan_ctx *ctx, stan_matrix *K) {
    if (K->d_vals48 || K->fx48_refused) return STAN_OK;
    if (K->nslots == 0) return STAN_OK;
    uint32_t *out;
    STANCHK(stan_dmalloc_streamed(ctx, (void **)&out, (size_t)K->nslots * 14 * 64 * 4,
                                  [&](const void *q, float *ms, bool self) {
                                      return stan_spmv_probe(ctx, K, q, (size_t)K->nslots * 14 * 64 * 4, STAN_PREC_FIXED48, ms, self);
                                  }));
    unsigned long long *d_bad = (unsigned long long *)(ctx->d_status + SS_COUNTER);
    HIPCHK(ctx, hipMemsetAsync(d_bad, 0, 8, ctx->stream));
    hipLaunchKernelGGL(k_to_fx48, dim3((unsigned)nblk(K->nslots * 64, 256)), dim3(256), 0, ctx->stream,
                       K->nslots, K->d_vals, out, d_bad);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_status + SS_COUNTER, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->h_status[SS_COUNTER] != 0) { stan_dfree(ctx, out); K->fx48_refused = true; return STAN_OK; }
    K->d_vals48 = out;
    return STAN_OK;
}

// Packed column stream of K (struct colstream), built once per matrix; the int32 columns stay (the
// assembly, scaling and export kernels use them).
int stan_matrix_make_cols16(stan_ctx *ctx, stan_matrix *K) {
    if (K->d_cols16 || K->nslices <= 0) return STAN_OK;
    return stan_pack_columns(ctx, K->nslices, K->nslots, K->d_slot_ptr, K->d_cols, &K->d_cols16, &K->d_colbase, &K->d_pair_ptr,
                             &K->d_slice_packed, &K->slots_packed, K->nloc, K->d_rowof, K->d_rowlen, &K->slots_packed2);
}
// the same for any sliced column stream (the folded copy of fold.hip has its own); *packed stays nullptr when the
// pair index would not fit an int32
int stan_pack_columns(stan_ctx *ctx, int32_t nslices, int64_t nslots, const int32_t *d_slot_ptr, const int32_t *d_cols,
                      uint32_t **packed_out, int32_t **base_out, int32_t **pair_ptr_out, uint8_t **ok_out, int64_t *slots_packed,
                      int64_t nloc, const int32_t *d_rowof, const int32_t *d_rowlen, int64_t *slots_packed2) {
    hipStream_t st_ = ctx->stream;
    dev_bufs bufs;
    int32_t *cnt; int64_t *ptr64;
    STANCHK(alloc(ctx, bufs, &cnt, (size_t)nslices + 1));
    STANCHK(alloc(ctx, bufs, &ptr64, (size_t)nslices + 2));
    hipLaunchKernelGGL(k_pair_counts, dim3(nblk(nslices, 256)), dim3(256), 0, st_, nslices, d_slot_ptr, cnt);
    STANCHK(stan_scan_exclusive(ctx, cnt, ptr64, nslices));
    std::vector<int64_t> h((size_t)nslices + 1);
    HIPCHK(ctx, hipMemcpyAsync(h.data(), ptr64, h.size() * 8, hipMemcpyDeviceToHost, st_));
    HIPCHK(ctx, hipStreamSynchronize(st_));
    const int64_t npairs = h[(size_t)nslices];
    if (npairs >= ((int64_t)1 << 31)) return STAN_OK;   // pair index is int32: keep the plain columns
    std::vector<int32_t> h32(h.size());
    for (size_t i = 0; i < h.size(); i++) h32[i] = (int32_t)h[i];
    STANCHK(stan_dmalloc(ctx, pair_ptr_out, h32.size()));
    // one allocation: [n] base, [n] base2, [nslices] cmask (64-bit words), n = max(nslots, 1): make_colstream
    const size_t nb_ = (size_t)(nslots > 0 ? nslots : 1);
    STANCHK(stan_dmalloc(ctx, base_out, 2 * nb_ + 2 * (size_t)nslices + 2));
    STANCHK(stan_dmalloc(ctx, ok_out, (size_t)nslices));
    uint32_t *packed;
    STANCHK(stan_dmalloc(ctx, &packed, (size_t)(npairs > 0 ? npairs : 1) * 64));
    HIPCHK(ctx, hipMemcpyAsync(*pair_ptr_out, h32.data(), h32.size() * 4, hipMemcpyHostToDevice, st_));
    int32_t *b2_ = *base_out + nb_;
    unsigned long long *cm_ = (unsigned long long *)(*base_out + 2 * nb_);
    if (((uintptr_t)cm_ & 7) != 0) cm_ = (unsigned long long *)((uintptr_t)cm_ + 4);   // (never: 2 n ints from an aligned block)
    hipLaunchKernelGGL(k_pack_cols, dim3(nblk(nslices, 4)), dim3(256), 0, st_, nslices, nloc, d_slot_ptr, d_cols, d_rowof, d_rowlen,
                       *pair_ptr_out, packed, *base_out, b2_, cm_, *ok_out);
    unsigned long long *d_cnt = (unsigned long long *)(ctx->d_status + SS_COUNTER);   // two words: SS_COUNTER, SS_H_ERRCOPY
    HIPCHK(ctx, hipMemsetAsync(d_cnt, 0, 16, st_));
    hipLaunchKernelGGL(k_count_ok, dim3(nblk(nslices, 256)), dim3(256), 0, st_, nslices, *ok_out, d_slot_ptr, d_cnt);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_status + SS_COUNTER, d_cnt, 16, hipMemcpyDeviceToHost, st_));
    HIPCHK(ctx, hipStreamSynchronize(st_));   // h32 must outlive the copy
    *slots_packed = ctx->h_status[SS_COUNTER];
    if (slots_packed2) *slots_packed2 = ctx->h_status[SS_COUNTER + 1];
    *packed_out = packed;
    return STAN_OK;
}

// NOTE on the halo layout: vectors that are gathered by the SpMV (p, x) hold the owned
// block rows first, padded to whole slices, then the halo block columns:
//   [ 3*nslices*64 owned+pad | 3*nhalo ]
// Local column indices >= nloc written by the symbolic phase are relative to nloc, so
// the halo region must start at 3*nloc: the pad only exists for nranks == 1 tails, where
// nhalo == 0.  For nranks > 1 every rank's row count is a multiple of 64 except the last
// rank's; the gather vectors are therefore sized 3*(max(nloc, pad) + nhalo) and the halo
// always sits at 3*nloc.
// Host waits of the loop.  Without peer-to-peer exchanges: the plain blocking calls.  With them the stream may sit in a
// wait for a peer that is gone, and a blocking call would never return (the polling wavefront keeps the queue busy):
// the host polls instead, and after stan_p2p_stall_seconds() without completion releases this rank's waits
// (stan_p2p_release_own), lets the queue drain and reports STAN_E_COMM.  `ev` == nullptr: the whole stream.
static int cg_wait(stan_ctx *ctx, bool p2p, hipStream_t st, hipEvent_t ev) {
    if (!p2p) {
        if (ev) HIPCHK(ctx, hipEventSynchronize(ev));
        else HIPCHK(ctx, hipStreamSynchronize(st));
        return STAN_OK;
    }
    const double bound = stan_p2p_stall_seconds();
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    for (;;) {
        const hipError_t e = ev ? hipEventQuery(ev) : hipStreamQuery(st);
        if (e == hipSuccess) return STAN_OK;
        if (e != hipErrorNotReady) { ctx->err = std::string("cg: ") + hipGetErrorString(e); (void)hipGetLastError(); return STAN_E_HIP; }
        (void)hipGetLastError();
        if (++spins > 200) std::this_thread::sleep_for(std::chrono::microseconds(spins > 2000 ? 500 : 50));
        if (ctx->p2p->broken.load()) break;   // somebody else (run_all, another rank's release) gave up already
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > bound) break;
    }
    stan_p2p_release_own(ctx);
    (void)hipStreamSynchronize(st);   // the released waits pass: what was enqueued runs out
    (void)hipGetLastError();
    ctx->err = "cg: peer-to-peer exchange made no progress (a peer rank failed or never arrived); this rank's waits were released";
    return STAN_E_COMM;
}

int stan_cg_device(stan_ctx *ctx, stan_matrix *K, const double *d_F, double eps_f,
                   int32_t max_its, int32_t precision_mode, double *d_U, int32_t *term_out,
                   int32_t *iters_out, double *rel_res_out) {
    if (precision_mode != STAN_PREC_FP64 && precision_mode != STAN_PREC_MIXED &&
        precision_mode != STAN_PREC_FIXED48) {
        ctx->err = "cg_solve: unknown precision_mode";
        return STAN_E_UNSUPPORTED;
    }
    if (eps_f < 0 || max_its < 0) {
        ctx->err = "cg_solve: eps_f and max_its must be >= 0";
        return STAN_E_ARG;
    }
    if (eps_f == 0 && max_its == 0) eps_f = 1.0e-6;  // lincgsetcond
    hipStream_t st_ = ctx->stream;
    event_bag events;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ctx->profiling) {
        ev0 = events.make(); ev1 = events.make();
        hipEventRecord(ev0, st_);
    }
    const bool dist = ctx->comm != nullptr || ctx->nranks > 1;  // exchanges in the loop
    // peer to peer (one-process group handle, STAN_OPT_COMM_P2P): no RCCL call below this line
    const bool p2p = dist && ctx->comm_p2p && ctx->p2p != nullptr;
    // no hipFree while the peers' streams wait for this rank's future exchanges (stan_ctx::defer_frees)
    struct free_later { stan_ctx *c; bool on; ~free_later() { if (on) { stan_flush_deferred(c); stan_p2p_ipc_trim(c); } } } free_guard{ctx, p2p};
    if (p2p) ctx->defer_frees = true;
    STANCHK(stan_cg_workspace(ctx, K));   // the context's vectors (the placement search probed with them)
    if (p2p) {
        // my neighbours write their boundary rows straight into these vectors: tell them where they are
        const int64_t ns_ = 3 * ((int64_t)K->nslices * 64 + K->nhalo);
        if (!K->d_scale) STANCHK(stan_dmalloc(ctx, &K->d_scale, (size_t)ns_));
        double *const pub[5] = {ctx->ws.xb[0], ctx->ws.xb[1], ctx->ws.p, ctx->ws.r, K->d_scale};
        STANCHK(stan_p2p_publish_vectors(ctx, K, pub));
    }
    STANCHK(ensure_scaled(ctx, K));
    if (ctx->cols16) STANCHK(stan_matrix_make_cols16(ctx, K));
    if (precision_mode == STAN_PREC_MIXED) STANCHK(stan_matrix_make_fp32(ctx, K));
    if (precision_mode == STAN_PREC_FIXED48) STANCHK(stan_matrix_make_fx48(ctx, K));
    // the stream the products really read (FIXED-48 falls back to fp64 when K is not SPD-scalable)
    const int vs = precision_mode == STAN_PREC_FIXED48 ? (K->d_vals48 ? STAN_PREC_FIXED48 : STAN_PREC_FP64)
                                                       : precision_mode;
    if (!stan_small_system(ctx, K)) {
        const int rc_fold = stan_matrix_make_folded(ctx, K, vs);
        if (rc_fold == STAN_E_ALLOC) {   // an optimisation must not fail the solve: the padded streams serve
            stan_matrix_abandon_folding(ctx, K);
            ctx->err.clear();
        } else
            STANCHK(rc_fold);
    }
    const bool sr = ctx->cg_single_reduce;
    const bool foldr = ctx->cg_fold_reduce;

    const int64_t n3 = 3 * K->nloc;
    const int64_t npad = (int64_t)K->nslices * 64;
    const int64_t ng = 3 * ((npad > K->nloc + K->nhalo ? npad : K->nloc + K->nhalo));
    const int64_t dof0 = 3 * K->r0;
    dev_bufs bufs;
    double *xb[2], *p, *r, *v, *w, *bh, *partial, *sc, *sv = nullptr;
    int64_t *stt;
    unsigned long long *tick;
    xb[0] = ctx->ws.xb[0]; xb[1] = ctx->ws.xb[1]; p = ctx->ws.p; r = ctx->ws.r;
    v = ctx->ws.v; w = ctx->ws.w; bh = ctx->ws.bh;
    if (sr) sv = ctx->ws.sv;
    const unsigned spmv_blocks = stan_small_system(ctx, K) ? (unsigned)K->nslices : nblk(K->nslices, 4);
    const size_t npart = 2 * (size_t)(spmv_blocks > VEC_BLOCKS ? spmv_blocks : VEC_BLOCKS) + 16;
    STANCHK(alloc(ctx, bufs, &partial, npart));
    STANCHK(alloc(ctx, bufs, &sc, (size_t)S_NSCAL));
    STANCHK(alloc(ctx, bufs, &stt, (size_t)T_NSTAT));
    STANCHK(alloc(ctx, bufs, &tick, (size_t)(2 * FOLD_WORDS)));   // two ticket-counter sets
    HIPCHK(ctx, hipMemsetAsync(sc, 0, S_NSCAL * 8, st_));
    HIPCHK(ctx, hipMemsetAsync(tick, 0, 2 * FOLD_WORDS * 8, st_));
    HIPCHK(ctx, hipMemsetAsync(xb[0], 0, (size_t)ng * 8, st_));
    HIPCHK(ctx, hipMemsetAsync(xb[1], 0, (size_t)ng * 8, st_));
    HIPCHK(ctx, hipMemsetAsync(p, 0, (size_t)ng * 8, st_));
    if (sr) {
        HIPCHK(ctx, hipMemsetAsync(r, 0, (size_t)ng * 8, st_));
        HIPCHK(ctx, hipMemsetAsync(sv, 0, (size_t)n3 * 8, st_));
    }
    // folded reductions: counter A serves the products, counter B the vector kernels
    // Where the sums of a reduction go.  One rank, or RCCL: the local scalars (RCCL all-reduces them in
    // place).  Peer to peer: every rank's mailbox slot `slot`, columns j0.. (+ arrival count when this is
    // the producer that completes the exchange); the consumers then read through a red_src.
    const stan_p2p_dev *p2p_tab = p2p ? stan_p2p_table(ctx) : nullptr;
    auto p2p_to = [&](int j0, bool signal) {
        return p2p ? p2p_out{p2p_tab, stan_p2p_reduce_slot(ctx), j0, signal ? 1 : 0} : NO_P2P;
    };
    auto fold_to = [&](int which, double *out, p2p_out po = NO_P2P) {
        return fold_args{foldr ? tick + FOLD_WORDS * which : nullptr, 0, 0, out, po};
    };
    auto reduce_if_unfolded = [&](int np, int nv, double *out, p2p_out po = NO_P2P, int64_t k_ = -1) {
        if (foldr || np <= 0) return;
        const int64_t *sk = k_ >= 1 ? stt : nullptr;   // (k_init's sum is formed before the status exists)
        if (nv == 2) hipLaunchKernelGGL(k_reduce<2>, dim3(1), dim3(256), 0, st_, partial, np, out, po, sk, k_);
        else hipLaunchKernelGGL(k_reduce<1>, dim3(1), dim3(256), 0, st_, partial, np, out, po, sk, k_);
    };
    // One exchange point of the sharded loop: RCCL all-reduce of `count` scalars in place, or the stream
    // wait for every rank's arrival; returns where the consumers find the sums.
    std::vector<hipEvent_t> red_ev, halo_ev;   // profiling: events around the exchanges
    int64_t n_coll = 0, n_wait = 0;            // RCCL collectives / stream waits enqueued by the loop (profile)
    auto exchange_sums = [&](double *scalars, int count, red_src *rs) -> int {
        *rs = red_src{nullptr, 0, nullptr, 0};
        if (!dist) return STAN_OK;
        if (ctx->profiling) { red_ev.push_back(events.make()); hipEventRecord(red_ev.back(), st_); }
        int rc_ = STAN_OK;
        if (p2p) {
            *rs = red_src{stan_p2p_mailbox(ctx, stan_p2p_reduce_slot(ctx)), ctx->nranks, nullptr, 0};
            rc_ = stan_p2p_reduce_wait(ctx, &rs->ctr, &rs->want);   // (wait mode 2: the consumers poll rs->ctr themselves)
            if (!rs->ctr) n_wait++;
        } else {
            rc_ = stan_comm_allreduce_sum_f64(ctx, scalars, (size_t)count);
            n_coll++;
        }
        if (ctx->profiling) { red_ev.push_back(events.make()); hipEventRecord(red_ev.back(), st_); }
        return rc_;
    };
    auto halo = [&](double *x) -> int {
        if (ctx->profiling) { halo_ev.push_back(events.make()); hipEventRecord(halo_ev.back(), st_); }
        const int rc_ = stan_comm_halo_exchange(ctx, K, x);
        if (p2p && !K->nbr.empty()) n_wait++;
        if (ctx->profiling) { halo_ev.push_back(events.make()); hipEventRecord(halo_ev.back(), st_); }
        return rc_;
    };

    const unsigned vg = vec_grid(n3);
    {
        fold_args f = fold_to(1, sc + S_VMV, p2p_to(0, true));
        f.nblocks = vg; f.np = (int)vg;
        hipLaunchKernelGGL(k_init, dim3(vg), dim3(VEC_T), 0, st_, n3, dof0, K->d_red, d_F, K->d_scale,
                           bh, xb[0], r, p, partial, f);
        reduce_if_unfolded((int)vg, 1, sc + S_VMV, p2p_to(0, true));
    }
    red_src rs_b;
    STANCHK(exchange_sums(sc + S_VMV, 1, &rs_b));
    hipLaunchKernelGGL(k_init_scalars, dim3(1), dim3(64), 0, st_, sc, stt, eps_f, rs_b);
    HIPCHK(ctx, hipGetLastError());

    // lincgcreate: ItsBeforeRestart = N (global reduced size)
    int64_t its_before_restart = K->n_red > 0 ? K->n_red : 1;
    const int64_t hard_cap = 0x7fffffff;  // iteration counter is int32 in the report

    // Sharded SpMV with the halo exchange hidden behind the interior slices: the slices whose
    // rows reference no halo column run on a side stream while the main stream packs, sends
    // and receives; the boundary slices follow on the main stream.  RCCL only ever sees the
    // main stream.
    const bool split = dist && ctx->overlap_halo && K->d_sl_bnd != nullptr;
    if (split && !ctx->side) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_a, hipEventDisableTiming));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_b, hipEventDisableTiming));
    }
    std::vector<hipEvent_t> spmv_ev, spmv2_ev;
    std::vector<int64_t> spmv_k, spmv2_k;  // iteration of each timed launch (see the profile below)
    int64_t n_launch = 0;                  // kernels enqueued by the loop (profile)
    // y = A^ x (x gets its halo filled first when sharded) with `dot` sums (k_spmv's DOT) reduced
    // into out[0..dot): folded into the last launch of the product, or by k_reduce.
    auto spmv = [&](double *x, double *y, int dot, double *out, int64_t k, p2p_out po = NO_P2P) -> int {
        if (ctx->profiling) {
            hipEvent_t a = events.make(), b = events.make();
            hipEventRecord(a, st_);
            spmv_ev.push_back(a); spmv_ev.push_back(b);
            spmv_k.push_back(k);
        }
        auto go = [&](int which, hipStream_t s, bool last) -> unsigned {
            const fold_args f = (dot && last) ? fold_to(0, out, po) : NO_FOLD;
            n_launch++;
            return dot == 2 ? launch_spmv_any<2>(ctx, K, vs, x, y, partial, stt, k, which, s, f)
                 : dot == 1 ? launch_spmv_any<1>(ctx, K, vs, x, y, partial, stt, k, which, s, f)
                            : launch_spmv_any<0>(ctx, K, vs, x, y, partial, stt, k, which, s, f);
        };
        unsigned parts = 0;
        bool folded = foldr;
        if (split) {
            HIPCHK(ctx, hipEventRecord(ctx->ev_a, st_));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_a, 0));
            parts = go(1, ctx->side, false);
            HIPCHK(ctx, hipEventRecord(ctx->ev_b, ctx->side));
            STANCHK(halo(x));
            HIPCHK(ctx, hipStreamWaitEvent(st_, ctx->ev_b, 0));
            const unsigned pb = go(2, st_, true);   // adds up the interior launch's partials too
            if (pb == 0) folded = false;            // no boundary slices on this rank: nobody folded
            parts += pb;
        } else {
            if (dist) STANCHK(halo(x));
            parts = go(0, st_, true);
            if (parts == 0) folded = false;
        }
        if (dot && !folded) {
            if (parts > 0 || po.pp) {   // (peer to peer: a rank that owns no rows still sends its zeros)
                const int64_t *sk = k >= 1 ? stt : nullptr;
                if (dot == 2) hipLaunchKernelGGL(k_reduce<2>, dim3(1), dim3(256), 0, st_, partial, (int)parts, out, po, sk, k);
                else hipLaunchKernelGGL(k_reduce<1>, dim3(1), dim3(256), 0, st_, partial, (int)parts, out, po, sk, k);
                n_launch++;
            } else HIPCHK(ctx, hipMemsetAsync(out, 0, 8 * dot, st_));   // a rank that owns no rows
        }
        if (ctx->profiling) hipEventRecord(spmv_ev.back(), st_);
        return STAN_OK;
    };

    // v = A^ x and w = A^ x2 in one matrix pass (fused residual refresh), x.v -> out
    auto spmv2 = [&](double *x, double *x2, double *out, int64_t k, p2p_out po = NO_P2P) -> int {
        if (ctx->profiling) {
            hipEvent_t a = events.make(), b = events.make();
            hipEventRecord(a, st_);
            spmv2_ev.push_back(a); spmv2_ev.push_back(b);
            spmv2_k.push_back(k);
        }
        auto go = [&](int which, hipStream_t s, bool last) -> unsigned {
            const fold_args f = last ? fold_to(0, out, po) : NO_FOLD;
            n_launch++;
            if (vs == STAN_PREC_FIXED48) return launch_spmv2<uint32_t>(ctx, K, K->d_vals48, x, x2, v, w, partial, stt, k, which, s, f);
            return vs == STAN_PREC_MIXED ? launch_spmv2<float>(ctx, K, K->d_vals32, x, x2, v, w, partial, stt, k, which, s, f)
                                         : launch_spmv2<double>(ctx, K, K->d_vals, x, x2, v, w, partial, stt, k, which, s, f);
        };
        unsigned parts = 0;
        bool folded = foldr;
        if (split) {
            HIPCHK(ctx, hipEventRecord(ctx->ev_a, st_));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_a, 0));
            parts = go(1, ctx->side, false);
            HIPCHK(ctx, hipEventRecord(ctx->ev_b, ctx->side));
            STANCHK(halo(x));
            STANCHK(halo(x2));
            HIPCHK(ctx, hipStreamWaitEvent(st_, ctx->ev_b, 0));
            const unsigned pb = go(2, st_, true);
            if (pb == 0) folded = false;
            parts += pb;
        } else {
            if (dist) { STANCHK(halo(x)); STANCHK(halo(x2)); }
            parts = go(0, st_, true);
            if (parts == 0) folded = false;
        }
        if (!folded) {
            if (parts > 0 || po.pp) { hipLaunchKernelGGL(k_reduce<1>, dim3(1), dim3(256), 0, st_, partial, (int)parts, out, po, (const int64_t *)stt, k); n_launch++; }
            else HIPCHK(ctx, hipMemsetAsync(out, 0, 8, st_));
        }
        if (ctx->profiling) hipEventRecord(spmv2_ev.back(), st_);
        return STAN_OK;
    };
    auto vec_fold = [&](double *out, p2p_out po = NO_P2P) {
        fold_args f = fold_to(1, out, po);
        f.nblocks = vg; f.np = (int)vg;
        return f;
    };

    int64_t *h_st = ctx->h_status + SS_H_CG_STATUS;  // pinned
    hipEvent_t poll[2] = {events.make(hipEventDisableTiming), events.make(hipEventDisableTiming)};
    int64_t k = 1;
    int chunk_id = 0;
    bool done = false;
    int rc = STAN_OK;
    // status of "iteration 0" (initial residual test)
    HIPCHK(ctx, hipMemcpyAsync(h_st, stt, T_NSTAT * 8, hipMemcpyDeviceToHost, st_));
    STANCHK(cg_wait(ctx, p2p, st_, nullptr));   // (peer to peer: behind the first reduction's wait)
    if (h_st[T_ITER_A] == 0) done = true;
    red_src rs_sr{nullptr, 0, nullptr, 0}, rs_vmv{nullptr, 0, nullptr, 0}, rs_r2{nullptr, 0, nullptr, 0};
    if (sr && !done) {   // w_0 = A r_0 with gamma_0, delta_0 (merit_0 = 0 sits in the zeroed scalars)
        if (p2p)         // ... or, peer to peer, is sent as this rank's zero into the slot of the first reduction
            hipLaunchKernelGGL(k_reduce<1>, dim3(1), dim3(256), 0, st_, partial, 0, sc + S_SR_MERIT, p2p_to(2, false), (const int64_t *)nullptr, (int64_t)0);
        rc = spmv(r, w, 2, sc + S_SR_GAMMA, 0, p2p_to(0, true));
        if (rc == STAN_OK) rc = exchange_sums(sc + S_SR_GAMMA, 3, &rs_sr);
    }
    // a sharded loop polls more often: what runs ahead of the stop are exchanges nobody can cut short
    const int chunk = dist ? CHUNK_DIST : CHUNK;
    while (!done && rc == STAN_OK) {
        // enqueue one chunk of iterations
        for (int c = 0; c < chunk && k < hard_cap; c++, k++) {
            const bool refresh = ctx->cg_rupdate > 0 && (k % ctx->cg_rupdate) == 0;
            if (sr) {
                sr_args a;
                a.n3 = n3; a.k = k; a.sc = sc; a.st = stt; a.epsf = eps_f; a.maxits = max_its;
                a.its_before_restart = its_before_restart; a.merit_stop = ctx->cg_merit_stop ? 1 : 0;
                a.refresh = refresh ? 1 : 0;
                a.xcur = xb[(k - 1) & 1]; a.xnext = xb[k & 1];
                a.r = r; a.p = p; a.s = sv; a.w = w; a.bh = bh; a.partial = partial;
                a.rs = rs_sr;
                // the merit sum rides in the third column of the coming reduction's mailbox slot (no count of its own)
                a.fold = vec_fold(sc + S_SR_MERIT, p2p_to(2, false));
                hipLaunchKernelGGL(k_vec_sr, dim3(vg), dim3(VEC_T), 0, st_, a);
                n_launch++;
                if (!refresh) reduce_if_unfolded((int)vg, 1, sc + S_SR_MERIT, p2p_to(2, false), k);
                else {   // r' = b^ - A^ x' (ALGLIB's periodic residual recomputation), then as usual
                    rc = spmv(xb[k & 1], v, 0, nullptr, k);
                    if (rc) break;
                    hipLaunchKernelGGL(k_refresh, dim3(vg), dim3(VEC_T), 0, st_, n3, k, (const int64_t *)stt,
                                       bh, v, xb[k & 1], r, partial, vec_fold(sc + S_SR_DELTA, p2p_to(1, false)));
                    n_launch++;
                    reduce_if_unfolded((int)vg, 2, sc + S_SR_DELTA, p2p_to(1, false), k);   // [r.r (rewritten below), merit]
                }
                rc = spmv(r, w, 2, sc + S_SR_GAMMA, k, p2p_to(0, true));
                if (rc) break;
                rc = exchange_sums(sc + S_SR_GAMMA, 3, &rs_sr);
                if (rc) break;
                continue;
            }
            const bool fused = refresh && ctx->cg_fused_refresh;
            rc = fused ? spmv2(p, xb[(k - 1) & 1], sc + S_VMV, k, p2p_to(0, true))
                       : spmv(p, v, 1, sc + S_VMV, k, p2p_to(0, true));
            if (rc) break;
            rc = exchange_sums(sc + S_VMV, 1, &rs_vmv);
            if (rc) break;
            const p2p_out po_r = p2p_to(0, true);   // the slot of r.r / merit (the wait above moved on to it)
            step_args a;
            a.rs_vmv = rs_vmv;
            a.n3 = n3; a.k = k; a.sc = sc; a.st = stt;
            a.xcur = xb[(k - 1) & 1]; a.xnext = xb[k & 1];
            a.r = r; a.p = p; a.v = v; a.w = w; a.bh = bh; a.partial = partial;
            a.refresh = refresh ? (fused ? 2 : 1) : 0;
            a.merit = ctx->cg_merit_stop ? 1 : 0;
            // x' = x + a p moves into k_update (p is read once for both updates: -79 MB of 714 per
            // iteration at 148^3) unless the merit sum needs x' here or a literal refresh multiplies it
            a.defer_x = (ctx->cg_defer_x && !a.merit && a.refresh != 1) ? 1 : 0;
            a.fold = vec_fold(sc + S_R2NEW, a.refresh == 1 ? NO_P2P : po_r);   // refresh 1: k_refresh forms the sums
            if (ctx->vec_store_nt & 2) hipLaunchKernelGGL(k_step<true>, dim3(vg), dim3(VEC_T), 0, st_, a);
            else hipLaunchKernelGGL(k_step<false>, dim3(vg), dim3(VEC_T), 0, st_, a);
            n_launch++;
            if (a.refresh == 1) {
                // a -5/-4 stop of this iteration is caught by k_refresh/k_update (ITER_B <= k)
                rc = spmv(xb[k & 1], v, 0, nullptr, k);
                if (rc) break;
                hipLaunchKernelGGL(k_refresh, dim3(vg), dim3(VEC_T), 0, st_, n3, k,
                                   (const int64_t *)stt, bh, v, xb[k & 1], r, partial, vec_fold(sc + S_R2NEW, po_r));
                n_launch++;
            }
            if (!foldr) { reduce_if_unfolded((int)vg, 2, sc + S_R2NEW, po_r, k); n_launch++; }
            rc = exchange_sums(sc + S_R2NEW, 2, &rs_r2);
            if (rc) break;
            const double *ux = a.defer_x ? a.xcur : nullptr;
            double *uxn = a.defer_x ? a.xnext : nullptr;
            if (ctx->vec_store_nt & 1)
                hipLaunchKernelGGL(k_update<true>, dim3(vg), dim3(VEC_T), 0, st_, n3, k, sc, stt, eps_f,
                                   (int64_t)max_its, its_before_restart, ctx->cg_merit_stop ? 1 : 0, r, p, ux, uxn, rs_r2, rs_vmv);
            else
                hipLaunchKernelGGL(k_update<false>, dim3(vg), dim3(VEC_T), 0, st_, n3, k, sc, stt, eps_f,
                                   (int64_t)max_its, its_before_restart, ctx->cg_merit_stop ? 1 : 0, r, p, ux, uxn, rs_r2, rs_vmv);
            n_launch++;
        }
        if (rc) break;
        // poll: read the status of the PREVIOUS chunk while this one runs
        int64_t *slot = h_st + 8 * (chunk_id & 1);
        HIPCHK(ctx, hipMemcpyAsync(slot, stt, T_NSTAT * 8, hipMemcpyDeviceToHost, st_));
        HIPCHK(ctx, hipEventRecord(poll[chunk_id & 1], st_));
        if (chunk_id > 0) {
            STANCHK(cg_wait(ctx, p2p, st_, poll[(chunk_id - 1) & 1]));
            int64_t *prev = h_st + 8 * ((chunk_id - 1) & 1);
            if (prev[T_TYPE] != 0) done = true;
        }
        if (k >= hard_cap) done = true;
        if (dist && (ctx->comm_broken.load() || (ctx->p2p && ctx->p2p->broken.load()))) {
            ctx->err = "cg: a peer rank failed (the exchanges were aborted)";
            rc = STAN_E_COMM;
        }
        chunk_id++;
    }
    if (p2p) {   // never a blocking wait on a stream that may sit in front of a peer that is gone
        if (rc == STAN_OK) rc = cg_wait(ctx, true, st_, nullptr);
        else stan_p2p_release_own(ctx);
    }
    hipError_t e = hipStreamSynchronize(st_);
    if (rc) return rc;
    if (e != hipSuccess) { ctx->err = std::string("cg: ") + hipGetErrorString(e); return STAN_E_HIP; }
    HIPCHK(ctx, hipMemcpyAsync(h_st, stt, T_NSTAT * 8, hipMemcpyDeviceToHost, st_));
    double *h_sc = (double *)(ctx->h_status + SS_H_CG_SCALARS);
    HIPCHK(ctx, hipMemcpyAsync(h_sc, sc, 16 * 8, hipMemcpyDeviceToHost, st_));
    HIPCHK(ctx, hipStreamSynchronize(st_));
    int type = (int)h_st[T_TYPE];
    int64_t its = h_st[T_ITERS];
    if (type == 0) { type = 5; its = k - 1; h_st[T_XSEL] = (k - 1) & 1; }  // hard cap
    const double *xfin = xb[h_st[T_XSEL] & 1];

    // U = S x^ on the free DOFs (a rank of a one-process group leaves only ITS entries [u0, u1) of U: the
    // group copies every rank's segment into the caller's buffer, nothing is gathered on the devices)
    if (!dist || ctx->result_segment) {
        hipLaunchKernelGGL(k_result, dim3(vg), dim3(VEC_T), 0, st_, n3, dof0, K->d_red, K->d_scale,
                           xfin, d_U);
    } else {
        double *full;
        STANCHK(alloc(ctx, bufs, &full, (size_t)K->n_dof));
        hipLaunchKernelGGL(k_result_full, dim3(vg), dim3(VEC_T), 0, st_, n3, K->d_scale, xfin,
                           full + dof0);
        STANCHK(stan_comm_allgather_rows(ctx, K, full));
        hipLaunchKernelGGL(k_compress, dim3(vec_grid(K->n_dof)), dim3(VEC_T), 0, st_, K->n_dof,
                           K->d_red, full, d_U);
    }
    HIPCHK(ctx, hipGetLastError());
    if (ctx->profiling) hipEventRecord(ev1, st_);
    HIPCHK(ctx, hipStreamSynchronize(st_));

    if (term_out) *term_out = type;
    if (iters_out) *iters_out = (int32_t)its;
    if (rel_res_out) *rel_res_out = h_sc[S_BNORM] > 0 ? std::sqrt(h_sc[S_R2OUT]) / h_sc[S_BNORM] : 0.0;
    if (ctx->profiling) {
        float ms = 0;
        hipEventElapsedTime(&ms, ev0, ev1);
        ctx->prof.cg_ms = ms;
        // Only launches that did work count: the host runs up to two chunks ahead of the status
        // it polls, so a converged solve is followed by a few dozen launches that return at once
        // (3-4 us each); averaging those in made the SpMV look ~3 % faster than it is.
        double tot = 0;
        int64_t nwork = 0;
        for (size_t i = 0; i + 1 < spmv_ev.size(); i += 2) {
            if (spmv_k[i / 2] > its) continue;
            float t = 0;
            hipEventElapsedTime(&t, spmv_ev[i], spmv_ev[i + 1]);
            tot += t;
            nwork++;
        }
        ctx->prof.spmv_ms_total = tot;
        ctx->prof.spmv_launches = nwork;
        double tot2 = 0;
        int64_t nwork2 = 0;
        for (size_t i = 0; i + 1 < spmv2_ev.size(); i += 2) {
            if (spmv2_k[i / 2] > its) continue;
            float t = 0;
            hipEventElapsedTime(&t, spmv2_ev[i], spmv2_ev[i + 1]);
            tot2 += t;
            nwork2++;
        }
        ctx->prof.spmv2_ms_total = tot2;
        ctx->prof.spmv2_launches = nwork2;
        ctx->prof.iterations = (int32_t)its;
        ctx->prof.termination_type = type;
        const int64_t blk_bytes = vs == STAN_PREC_FIXED48 ? 60 : vs == STAN_PREC_MIXED ? 40 : 76;
        // bytes of the format actually streamed: a block of a packed slice carries a 2-B column offset
        // instead of a 4-B index (+ 4 B per slot for its base, shared by 64 rows)
        const bool packed = ctx->cols16 && K->d_cols16;
        const double packed_frac = packed && K->nslots > 0 ? (double)K->slots_packed / (double)K->nslots : 0.0;
        ctx->prof.spmv_bytes = K->nblocks * blk_bytes + 3 * K->nloc * 16 + K->nloc * 4
                               - (int64_t)(packed_frac * (double)K->nblocks * 2.0) + (packed ? (K->slots_packed + K->slots_packed2) * 4 : 0);
        const bool folded = ctx->row_folding != 0 && (vs == STAN_PREC_FIXED48 ? K->d_fold_vals48 != nullptr : vs == STAN_PREC_MIXED ? K->d_fold_vals32 != nullptr
                                                                                                                                    : K->d_fold_vals != nullptr);
        if (folded) {   // its own packed column stream, 4 B of plan per row
            const bool fp = ctx->cols16 && K->d_fold_cols16;
            const double ff = fp && K->nfslots > 0 ? (double)K->fold_slots_packed / (double)K->nfslots : 0.0;
            ctx->prof.spmv_bytes = K->nblocks * blk_bytes + 3 * K->nloc * 16 + K->nloc * 8 - (int64_t)(ff * (double)K->nblocks * 2.0) +
                                   (fp ? K->fold_slots_packed * 4 : 0);
            ctx->prof.col_slots_packed = fp ? K->fold_slots_packed : 0;
        }
        ctx->prof.repacked_streams = folded ? 1 : 0;
        ctx->prof.col_slots_packed = packed ? K->slots_packed : 0;
        ctx->prof.value_stream = vs;
        // vector passes of one classic iteration: k_step reads r, v (+ p, x unless deferred; + b^ for the
        // merit sum) and writes r (+ x); k_update reads r, p (+ x when deferred) and writes p (+ x)
        ctx->prof.cg_iteration_vector_bytes = 3 * K->nloc * 8 * ((ctx->cg_defer_x && !ctx->cg_merit_stop ? 8 : 9) + (ctx->cg_merit_stop ? 1 : 0));
        ctx->prof.loop_kernel_launches = n_launch;
        ctx->prof.loop_collectives = n_coll;
        ctx->prof.loop_stream_waits = n_wait;
        ctx->prof.loop_iterations_enqueued = k - 1;
        // what the stream spent in the exchanges (RCCL launches, or peer-to-peer waits): events around each
        auto sum_pairs = [](const std::vector<hipEvent_t> &ev, double *tot, int64_t *cnt) {
            *tot = 0; *cnt = 0;
            for (size_t i = 0; i + 1 < ev.size(); i += 2) {
                float t = 0;
                if (hipEventElapsedTime(&t, ev[i], ev[i + 1]) == hipSuccess) { *tot += t; (*cnt)++; }
            }
        };
        sum_pairs(red_ev, &ctx->prof.comm_reduce_ms_total, &ctx->prof.comm_reduce_calls);
        sum_pairs(halo_ev, &ctx->prof.comm_halo_ms_total, &ctx->prof.comm_halo_calls);
    }
    return STAN_OK;
}

// y = K x on the reduced system (test helper; single rank)
int stan_spmv_reduced(stan_ctx *ctx, stan_matrix *K, const double *d_x, double *d_y) {
    if (ctx->nranks != 1) { ctx->err = "spmv: single-rank contexts only"; return STAN_E_UNSUPPORTED; }
    hipStream_t st_ = ctx->stream;
    const int64_t n3 = 3 * K->nloc, npad3 = 3 * (int64_t)K->nslices * 64;
    dev_bufs bufs;
    double *xf, *yf; int64_t *stt;
    STANCHK(alloc(ctx, bufs, &xf, (size_t)npad3));
    STANCHK(alloc(ctx, bufs, &yf, (size_t)npad3));
    STANCHK(alloc(ctx, bufs, &stt, (size_t)T_NSTAT));
    int64_t init[T_NSTAT] = {0x7fffffffffffffffLL, 0x7fffffffffffffffLL, 0, 0, 0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(stt, init, sizeof(init), hipMemcpyHostToDevice, st_));
    HIPCHK(ctx, hipMemsetAsync(xf, 0, (size_t)npad3 * 8, st_));
    const double *sdiv = K->scaled ? K->d_scale : nullptr;
    // K x = S^-1 (A^ (S^-1 x)) when the matrix already carries its scaling
    hipLaunchKernelGGL(k_expand, dim3(vec_grid(n3)), dim3(VEC_T), 0, st_, n3, (int64_t)0, K->d_red,
                       d_x, sdiv, xf);
    launch_spmv<double, 0>(ctx, K, K->d_vals, xf, yf, nullptr, stt, 1);
    // compress (and undo the row scaling)
    hipLaunchKernelGGL(k_compress_div, dim3(vec_grid(n3)), dim3(VEC_T), 0, st_, n3, K->d_red, sdiv, yf, d_y);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(st_));
    return STAN_OK;
}

// y_owned = A_local x_local, x_local = [owned rows | halo columns] (plan checks; any rank)
int stan_spmv_local(stan_ctx *ctx, stan_matrix *K, const double *d_x, double *d_y) {
    dev_bufs bufs;
    int64_t *stt;
    STANCHK(alloc(ctx, bufs, &stt, (size_t)T_NSTAT));
    int64_t init[T_NSTAT] = {0x7fffffffffffffffLL, 0x7fffffffffffffffLL, 0, 0, 0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(stt, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    if (K->d_sl_bnd) {  // sharded: interior + boundary lists must cover every slice exactly once
        HIPCHK(ctx, hipMemsetAsync(d_y, 0xff, (size_t)(3 * K->nloc) * 8, ctx->stream));  // NaN
        launch_spmv<double, 0>(ctx, K, K->d_vals, d_x, d_y, nullptr, stt, 1, 1);
        launch_spmv<double, 0>(ctx, K, K->d_vals, d_x, d_y, nullptr, stt, 1, 2);
    } else
        launch_spmv<double, 0>(ctx, K, K->d_vals, d_x, d_y, nullptr, stt, 1);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return STAN_OK;
}

int stan_spmv_bench_device(stan_ctx *ctx, stan_matrix *K, int32_t precision_mode, int32_t reps,
                           double *avg_ms) {
    hipStream_t st_ = ctx->stream;
    const bool mixed = precision_mode == STAN_PREC_MIXED;
    if (mixed) STANCHK(stan_matrix_make_fp32(ctx, K));
    if (precision_mode == STAN_PREC_FIXED48) {
        STANCHK(ensure_scaled(ctx, K));
        STANCHK(stan_matrix_make_fx48(ctx, K));
        if (!K->d_vals48) { ctx->err = "spmv_bench: matrix not representable in FIXED48"; return STAN_E_UNSUPPORTED; }
    }
    const bool fx = precision_mode == STAN_PREC_FIXED48;
    const int64_t npad = (int64_t)K->nslices * 64;
    const int64_t ng = 3 * ((npad > K->nloc + K->nhalo ? npad : K->nloc + K->nhalo));
    dev_bufs bufs;
    double *x, *y, *partial; int64_t *stt;
    // the CG's own gather vector and product buffer: the pair (value block, vector blocks) that is
    // timed here is the pair the solve will run on
    STANCHK(stan_cg_workspace(ctx, K));
    x = ctx->ws.p; y = ctx->ws.v;
    STANCHK(alloc(ctx, bufs, &partial, 2 * (size_t)K->nslices + 2));   // k_spmv_small leaves one partial per slice
    STANCHK(alloc(ctx, bufs, &stt, (size_t)T_NSTAT));
    int64_t init[T_NSTAT] = {0x7fffffffffffffffLL, 0x7fffffffffffffffLL, 0, 0, 0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(stt, init, sizeof(init), hipMemcpyHostToDevice, st_));
    hipLaunchKernelGGL(k_fill, dim3(vec_grid(ng)), dim3(VEC_T), 0, st_, x, ng, 1.0);
    event_bag events;
    hipEvent_t a = events.make(), b = events.make();
    auto one = [&]() {
        if (mixed) launch_spmv<float, 1>(ctx, K, K->d_vals32, x, y, partial, stt, 1);
        else if (fx) launch_spmv<uint32_t, 1>(ctx, K, K->d_vals48, x, y, partial, stt, 1);
        else launch_spmv<double, 1>(ctx, K, K->d_vals, x, y, partial, stt, 1);
    };
    for (int i = 0; i < 3; i++) one();
    hipEventRecord(a, st_);
    for (int i = 0; i < reps; i++) one();
    hipEventRecord(b, st_);
    HIPCHK(ctx, hipEventSynchronize(b));
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    *avg_ms = reps > 0 ? ms / reps : 0;
    HIPCHK(ctx, hipGetLastError());
    return STAN_OK;
}

// Time of the fp64 SpMV of K streaming its values from `vals` (any contents: only the addresses
// matter), median of 3 launches after a warm-up.  Used by the allocation-by-trial of placement.hip.
// self_pair: the gather vector and the product are carved out of the FRONT of the candidate block
// itself instead of the context's vectors -- by construction the same-group (slow) pairing, i.e.
// the reference the search compares the real pairing with (profiles/r02/placement_cross_self_n148.txt).
int stan_spmv_probe(stan_ctx *ctx, stan_matrix *K, const void *vals, size_t bytes, int32_t precision,
                    float *ms_out, bool self_pair) {
    hipStream_t st_ = ctx->stream;
    *ms_out = 0;
    if (K->nslices <= 0) return STAN_OK;
    const int64_t npad = (int64_t)K->nslices * 64;
    const int64_t ng = 3 * ((npad > K->nloc + K->nhalo ? npad : K->nloc + K->nhalo));
    dev_bufs bufs;
    double *x, *y, *partial; int64_t *stt;
    // the CG's own gather vector and product buffer: the pair (value block, vector blocks) that is
    // timed here is the pair the solve will run on
    STANCHK(stan_cg_workspace(ctx, K));
    x = ctx->ws.p; y = ctx->ws.v;
    if (self_pair) {   // the block holds no values yet (only addresses matter to the timing)
        // the gather vector and the product must fit into the candidate: a stream with few slots per
        // slice (or a large halo) has no self-paired reference -- *ms_out stays 0, the search then
        // keeps the fastest real pairing (placement.hip)
        if ((size_t)(((ng + 511) & ~(int64_t)511) + 3 * K->nloc) * 8 > bytes) return STAN_OK;
        x = (double *)const_cast<void *>(vals);
        y = x + ((ng + 511) & ~(int64_t)511);
    }
    STANCHK(alloc(ctx, bufs, &partial, 2 * (size_t)K->nslices + 2));   // k_spmv_small leaves one partial per slice
    STANCHK(alloc(ctx, bufs, &stt, (size_t)T_NSTAT));
    int64_t init[T_NSTAT] = {0x7fffffffffffffffLL, 0x7fffffffffffffffLL, 0, 0, 0, 0, 0, 0};
    HIPCHK(ctx, hipMemcpyAsync(stt, init, sizeof(init), hipMemcpyHostToDevice, st_));
    hipLaunchKernelGGL(k_fill, dim3(vec_grid(ng)), dim3(VEC_T), 0, st_, x, ng, 1.0);
    event_bag ev;
    float t[3] = {0, 0, 0};
    for (int r = 0; r < 4; r++) {
        hipEvent_t a = ev.make(), b = ev.make();
        hipEventRecord(a, st_);
        if (precision == STAN_PREC_FIXED48)
            launch_spmv<uint32_t, 1>(ctx, K, (const uint32_t *)vals, x, y, partial, stt, 1);
        else if (precision == STAN_PREC_MIXED)
            launch_spmv<float, 1>(ctx, K, (const float *)vals, x, y, partial, stt, 1);
        else
            launch_spmv<double, 1>(ctx, K, (const double *)vals, x, y, partial, stt, 1);
        hipEventRecord(b, st_);
        HIPCHK(ctx, hipEventSynchronize(b));
        if (r > 0) hipEventElapsedTime(&t[r - 1], a, b);
    }
    HIPCHK(ctx, hipGetLastError());
    const float lo = t[0] < t[1] ? t[0] : t[1], hi = t[0] < t[1] ? t[1] : t[0];
    *ms_out = t[2] < lo ? lo : (t[2] > hi ? hi : t[2]);
    return STAN_OK;
}

#ifdef STAN_LAB
#include "lab/cg_lab.inc"   // lab-only host entry points
#endif

// un-scale on export
int stan_matrix_unscale(stan_ctx *ctx, stan_matrix *K) {
    if (!K->scaled) return STAN_OK;
    if (K->nslices > 0)
        hipLaunchKernelGGL(k_scale_matrix, dim3(nblk(K->nslices, 4)), dim3(256), 0, ctx->stream,
                           K->nslices, K->d_slot_ptr, K->d_rowof, K->d_cols, K->d_vals, K->d_scale, 1);
    HIPCHK(ctx, hipGetLastError());
    K->scaled = false;
    if (K->d_vals32) { stan_dfree(ctx, K->d_vals32); K->d_vals32 = nullptr; }
    if (K->d_vals48) { stan_dfree(ctx, K->d_vals48); K->d_vals48 = nullptr; }
    stan_matrix_drop_folded_values(ctx, K);
    K->fx48_refused = false;
    return STAN_OK;
}
