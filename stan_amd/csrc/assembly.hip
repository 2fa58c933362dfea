// assembly.hip -- HEX8 stiffness assembly on gfx950: replaces
// SolverFunctions.ParallelAssembly_K (SolverFunctions.cs:117-180) + alglib
// sparsecreate/sparseadd/sparseconverttocrs (:123,:164,:275).
//
// Data layout in HBM ("BSELL-64"): the global K is stored as 3x3 node blocks in
// sliced-ELL form.  Block row = node in reference DOF order (Node.DOF[0]/3,
// Database.cs:140-234).  A slice is 64 consecutive block rows (one wavefront);
// slice s owns the k-slots [slot_ptr[s], slot_ptr[s+1]); slot k of a slice holds,
// for each of its 64 rows, the k-th block of the row (columns ascending):
//     cols[slot][lane]            int32  local block-column index
//     vals[slot][comp 0..8][lane] double block entry (row-major 3x3)
// so a wavefront streaming a slice issues fully coalesced 512-B loads.
//
// Fixed DOFs (nDOF_reduction == -1, Solver.cs:121-132) are kept as identity
// rows/columns instead of being squeezed out: entries in a fixed row or column
// are zero, the fixed diagonal is 1.  With b = 0 on fixed DOFs the CG iterates on
// the free DOFs are those of the reduced system the reference builds
// (SolverFunctions.cs:155-165), and 3x3 blocks stay intact.
//
// Assembly is a row-owner GATHER, not a scatter: one wavefront owns one block row,
// walks the (element, local node) incidences of its node in ascending element order
// and sums the element blocks K_e[a][b] it recomputes from the nodal coordinates
// (lane = (incidence s, local node b); the eight lanes of an incidence share the
// element's J^-1 and c grad N_a at the 8 Gauss points through LDS).  Every value of K is
// written to memory exactly once, with no global atomics and a fixed summation order (the
// row's blocks add up in LDS, one incidence after the other: bit-reproducible, unlike the
// reference's lock(K) scatter, SolverFunctions.cs:162).
#include <algorithm>

#include "internal.h"
#include "hex8_device.h"

namespace {

constexpr int ERR_DOF_LAYOUT = 1, ERR_CONN_RANGE = 2;

// XCD-chunked workgroup mapping (the SpMV's, cg.hip).  The hardware deals consecutive workgroups round-robin to the 8
// XCDs, each with an L2 of its own; with the identity mapping every L2 sees every eighth workgroup of the whole sweep.
// Dealt in windows of 8 * ch, each XCD gets ch CONSECUTIVE logical workgroups, so its L2 holds one contiguous piece of
// the mesh at a time (the ragged tail keeps the identity mapping).
__device__ __forceinline__ int64_t xcd_chunked(int64_t bid, int64_t grid, int ch) {
    const int64_t win = 8 * (int64_t)ch, grp = bid / win, within = bid - grp * win;
    return (grp + 1) * win <= grid ? grp * win + (within & 7) * ch + (within >> 3) : bid;
}

// ---- step 0: node permutation + fixed-DOF masks ------------------------------------------
// (+ the coordinates in BLOCK-ROW order: the numeric phase gathers the nodes of a row's elements, which are
// neighbours in row order but strided all over the wire order)
__global__ void k_perm(int64_t n_nodes, int64_t nb_glob, const int32_t *node_dof,
                       const int32_t *red, const double *xyz, int32_t *perm, uint8_t *fixmask, double *xrow,
                       int64_t *status) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const int32_t d0 = node_dof[3 * i], d1 = node_dof[3 * i + 1], d2 = node_dof[3 * i + 2];
    // Node.cs:218-223 SetDOF: DOF = {3*index, 3*index+1, 3*index+2}
    if (d0 < 0 || d0 % 3 != 0 || d1 != d0 + 1 || d2 != d0 + 2 || d0 / 3 >= nb_glob) {
        atomicOr((unsigned long long *)&status[SS_ERRBITS], (unsigned long long)ERR_DOF_LAYOUT);
        perm[i] = 0;
        return;
    }
    const int32_t r = d0 / 3;
    perm[i] = r;
    xrow[3 * (int64_t)r] = xyz[3 * i]; xrow[3 * (int64_t)r + 1] = xyz[3 * i + 1]; xrow[3 * (int64_t)r + 2] = xyz[3 * i + 2];
    uint8_t m = 0;
    if (red[d0] == -1) m |= 1;
    if (red[d1] == -1) m |= 2;
    if (red[d2] == -1) m |= 4;
    fixmask[r] = m;
}

// ---- step 1: node -> (element, local node) incidence lists for owned rows -----------------
// (+ the connectivity as GLOBAL BLOCK ROWS, crow = perm[conn]: every later phase wants the row, and the
// numeric phase saves one dependent gather per incidence)
__global__ void k_count_incident(int64_t n_elem, int64_t n_nodes, const int32_t *conn,
                                 const int32_t *perm, int64_t r0, int64_t r1, int32_t *cnt,
                                 int32_t *crow, int64_t *status) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_elem * 8) return;
    const int32_t nd = conn[t];
    if (nd < 0 || nd >= n_nodes) {
        atomicOr((unsigned long long *)&status[SS_ERRBITS], (unsigned long long)ERR_CONN_RANGE);
        crow[t] = 0;
        return;
    }
    const int64_t row = perm[nd];
    crow[t] = (int32_t)row;
    if (row >= r0 && row < r1) atomicAdd(&cnt[row - r0], 1);
}

__global__ void k_fill_incident(int64_t n_elem, const int32_t *crow, int64_t r0, int64_t r1,
                                const int64_t *ptr, int32_t *cursor, int32_t *list) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_elem * 8) return;
    const int64_t row = crow[t];   // (a connectivity entry out of range has ended the call before this launch)
    if (row >= r0 && row < r1) {
        const int32_t pos = atomicAdd(&cursor[row - r0], 1);
        list[ptr[row - r0] + pos] = (int32_t)t;  // t = e*8 + a
    }
}

// ---- bitonic sort of P (power of two >= 64) ints in LDS by one 64-lane workgroup -----------
__device__ inline void lds_bitonic_sort(int32_t *a, int P) {
    const int lane = threadIdx.x;
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < P; i += 64) {
                const int l = i ^ j;
                if (l > i) {
                    const int32_t x = a[i], y = a[l];
                    const bool up = (i & k) == 0;
                    if (up ? (x > y) : (x < y)) {
                        a[i] = y;
                        a[l] = x;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// bitonic sort of one value per lane across the 64-lane wavefront, in registers
__device__ inline int32_t wave_bitonic_sort(int32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int32_t o = __shfl_xor(v, j, 64);
            const bool keepmin = ((lane & j) == 0) == ((lane & k) == 0);
            v = keepmin ? min(v, o) : max(v, o);
        }
    }
    return v;
}

// ---- step 2: symbolic.  One 64-lane workgroup per owned block row, ONE sort per row ---------
// Sorts the row's incidence list in place (ascending element index, then local node: the numeric
// phase accumulates in that order), sorts the candidate neighbour block rows, and leaves the
// distinct ones -- ascending GLOBAL index -- in ucols[8 * ptr[row] ...] for k_fill_cols; counts
// them (rowlen) and flags referenced non-owned block rows (halo discovery).  (Rounds 1-2 ran this
// kernel twice, count and fill, and sorted every row both times.)
__global__ void __launch_bounds__(64)
k_symbolic(int64_t nloc, int64_t r0, int64_t r1, const int64_t *ptr, int32_t *list,
           const int32_t *crow, int32_t *rowlen, int32_t *refflag,
           int32_t *ucols, int64_t *status) {
    __shared__ int32_t ent[64];
    __shared__ int32_t cand[8 * STAN_MAX_INCIDENT];
    const int lane = threadIdx.x;
    const int64_t row = xcd_chunked(blockIdx.x, gridDim.x, 256);  // local row (256 consecutive rows per XCD); rows >= nloc are slice padding
    if (row >= nloc) {
        if (lane == 0) rowlen[row] = 0;
        return;
    }
    const int64_t p0 = ptr[row];
    const int deg = (int)(ptr[row + 1] - p0);
    if (deg > STAN_MAX_INCIDENT) return;   // a high-valence node: k_symbolic_big (one workgroup, any number of incidences)
    // incidence entries ascending (= ascending element index, then local node)
    {
        int32_t en = lane < deg ? list[p0 + lane] : 0x7fffffff;
        en = wave_bitonic_sort(en);
        if (lane < deg) list[p0 + lane] = en;
        ent[lane] = en;
    }
    __syncthreads();
    const int ncand = deg * 8;
    int P = 64;
    while (P < ncand) P <<= 1;
    for (int i = lane; i < P; i += 64) {
        int32_t c = 0x7fffffff;
        if (i < ncand) {
            const int32_t e = ent[i >> 3] >> 3;
            c = crow[(int64_t)e * 8 + (i & 7)];
        }
        cand[i] = c;
    }
    __syncthreads();
    if (P == 64) {  // the common case (<= 8 incident elements): sort in registers
        cand[lane] = wave_bitonic_sort(cand[lane]);
        __syncthreads();
    } else
        lds_bitonic_sort(cand, P);
    // distinct values, in ascending (global) order
    int32_t *uc = ucols + 8 * p0;   // capacity 8 * deg >= number of distinct candidates
    int32_t base = 0;
    for (int i0 = 0; i0 < P; i0 += 64) {
        const int i = i0 + lane;
        const int32_t c = cand[i];
        const bool isnew = c != 0x7fffffff && (i == 0 || cand[i - 1] != c);
        const unsigned long long m = __ballot(isnew);
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (isnew) {
            uc[pos] = c;
            if (refflag && (c < r0 || c >= r1)) refflag[c] = 1;
        }
        base += __popcll(m);
    }
    if (lane == 0) rowlen[row] = base;   // (no limit: slices wider than the fast numeric kernel's LDS go to k_numeric_wide)
}

// ---- step 2, high-valence nodes.  The reference puts no bound on the elements at a node (Database.cs:149-176 builds
// the lists, SolverFunctions.cs:143-173 scatters whatever K_e it gets): a solid of revolution meshed with collapsed
// hexes in 5-degree sectors has 72 of them on its axis.  Rows with more than STAN_MAX_INCIDENT incidences are rare and
// take this slow path: one 256-thread workgroup per row, the same two sorts (incidences; candidate block rows) as a
// bitonic network over a power-of-two buffer -- in LDS when it fits the launch's allotment, otherwise in a slice of
// global scratch claimed with a ticket (same code: one workgroup, barriers between the steps).
__device__ inline void wg_bitonic_sort(int32_t *a, int P) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += blockDim.x) {
                const int l = i ^ j;
                if (l > i) {
                    const int32_t x = a[i], y = a[l];
                    const bool up = (i & k) == 0;
                    if (up ? (x > y) : (x < y)) { a[i] = y; a[l] = x; }
                }
            }
            __syncthreads();
        }
    }
}
__global__ void __launch_bounds__(256)
k_symbolic_big(const int32_t *big_rows, int64_t r0, int64_t r1, const int64_t *ptr, int32_t *list, const int32_t *crow,
               int32_t *rowlen, int32_t *refflag, int32_t *ucols, int lds_ints, int32_t *scratch, int64_t scratch_ints_per_row,
               unsigned long long *ticket) {
    extern __shared__ int32_t big_lds[];
    __shared__ int32_t *sh_buf;
    __shared__ int sh_base;
    const int64_t row = big_rows[blockIdx.x];   // the rows k_symbolic left out, in no particular order (k_list_big_rows)
    const int64_t p0 = ptr[row];
    const int64_t deg = ptr[row + 1] - p0;
    int64_t PD = 64, PC = 64;
    while (PD < deg) PD <<= 1;
    while (PC < 8 * deg) PC <<= 1;
    if (threadIdx.x == 0) {
        if (PD + PC <= (int64_t)lds_ints) sh_buf = big_lds;
        else sh_buf = scratch + (int64_t)atomicAdd(ticket, 1ULL) * scratch_ints_per_row;
    }
    __syncthreads();
    int32_t *ent = sh_buf, *cand = sh_buf + PD;
    for (int64_t i = threadIdx.x; i < PD; i += 256) ent[i] = i < deg ? list[p0 + i] : 0x7fffffff;
    __syncthreads();
    wg_bitonic_sort(ent, (int)PD);
    for (int64_t i = threadIdx.x; i < deg; i += 256) list[p0 + i] = ent[i];
    for (int64_t i = threadIdx.x; i < PC; i += 256)
        cand[i] = i < 8 * deg ? crow[(int64_t)(ent[i >> 3] >> 3) * 8 + (i & 7)] : 0x7fffffff;
    __syncthreads();
    wg_bitonic_sort(cand, (int)PC);
    // distinct values in ascending (global) order, 256 at a time: position = distinct values so far + those in front of me
    __shared__ int sh_cnt[4];
    int32_t *uc = ucols + 8 * p0;
    if (threadIdx.x == 0) sh_base = 0;
    __syncthreads();
    for (int64_t i0 = 0; i0 < PC; i0 += 256) {
        const int64_t i = i0 + threadIdx.x;
        const int32_t c = cand[i];
        const bool isnew = c != 0x7fffffff && (i == 0 || cand[i - 1] != c);
        const unsigned long long m = __ballot(isnew);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane == 0) sh_cnt[w] = __popcll(m);
        __syncthreads();
        int pos = sh_base + __popcll(m & ((1ull << lane) - 1ull));
        for (int q = 0; q < w; q++) pos += sh_cnt[q];
        if (isnew) {
            uc[pos] = c;
            if (refflag && (c < r0 || c >= r1)) refflag[c] = 1;
        }
        __syncthreads();
        if (threadIdx.x == 0) sh_base += sh_cnt[0] + sh_cnt[1] + sh_cnt[2] + sh_cnt[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) rowlen[row] = sh_base;
}

// most incidences of one owned row, and the rows whose sort buffers exceed `lds_ints` (they need global scratch)
__global__ void k_max_incident(int64_t nrows, const int32_t *cnt, int lds_ints, int64_t *status) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int v = i < nrows ? cnt[i] : 0;
    bool giant = false;
    if (v > STAN_MAX_INCIDENT) {
        int64_t PD = 64, PC = 64;
        while (PD < v) PD <<= 1;
        while (PC < 8 * (int64_t)v) PC <<= 1;
        giant = PD + PC > (int64_t)lds_ints;
    }
    int m = v;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
    const unsigned long long g = __ballot(giant), bg = __ballot(v > STAN_MAX_INCIDENT);
    if ((threadIdx.x & 63) == 0) {
        if (m > STAN_MAX_INCIDENT) atomicMax((long long *)&status[SS_MAXDEG], (long long)m);
        if (g) atomicAdd((unsigned long long *)&status[SS_NGIANT], (unsigned long long)__popcll(g));
        if (bg) atomicAdd((unsigned long long *)&status[SS_NBIG], (unsigned long long)__popcll(bg));
    }
}
// out[0 .. *counter) = the indices i with v[i] > thresh (any order: every entry is handled on its own)
__global__ void k_list_above(int64_t n, const int32_t *v, int thresh, int32_t *out, unsigned long long *counter) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && v[i] > thresh) out[atomicAdd(counter, 1ULL)] = (int32_t)i;
}

// ---- step 2b: SELL-C-sigma.  Rows sorted by length (descending, stable) inside windows of sigma
// slices: a slice is as wide as its longest row, so 64 rows of similar length waste no slots.
// On the regular cube a slice of reference-order rows is 1-2 % padding; on a box with 15 % / 40 %
// of its elements knocked out 8.3 % / 25.7 % (tools/sellcs_padding.py), every padded slot being
// streamed by the SpMV like a real one; sorted in windows of 32 slices: 0.9 % / 1.4 %.
// The permutation stays INSIDE the matrix: rowof[position] = local block row, posof = inverse;
// vectors, halo plan, CRS export and the row partition keep the reference (AssignDOF) order, and
// a row's blocks keep their ascending column order, so every row sum keeps its bits.
// One WAVEFRONT per window, a stable counting sort: histogram of the lengths (<= STAN_MAX_ROW_BLOCKS),
// start of every length class (longest first), then the window's rows in order, 64 at a time -- a row's
// place in its class = rows of that length seen so far + equal-length lanes in front of it in the wave.
__global__ void __launch_bounds__(64)
k_window_sort(int64_t npad, int sigma, const int32_t *rowlen, int32_t *rowof, int32_t *posof) {
    constexpr int NL = 128;            // length classes (row lengths are clipped into them: only the order suffers)
    __shared__ int32_t start[NL];      // histogram, then the running start of every class
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * sigma * 64;
    const int W = (int)((npad - base) < (int64_t)sigma * 64 ? (npad - base) : (int64_t)sigma * 64);
    for (int i = lane; i < NL; i += 64) start[i] = 0;
    __syncthreads();
    for (int i = lane; i < W; i += 64) {
        int l = rowlen[base + i];
        l = l < NL ? l : NL - 1;
        atomicAdd(&start[l], 1);
    }
    __syncthreads();
    {   // exclusive prefix over the classes, longest first (two classes per lane, serial: NL is tiny)
        int32_t h0 = start[NL - 1 - 2 * lane], h1 = start[NL - 2 - 2 * lane];
        int32_t incl = h0 + h1;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        const int32_t excl = incl - (h0 + h1);
        __syncthreads();
        start[NL - 1 - 2 * lane] = excl;
        start[NL - 2 - 2 * lane] = excl + h0;
    }
    __syncthreads();
    for (int c = 0; c < W; c += 64) {
        const int i = c + lane;
        int l = i < W ? rowlen[base + i] : -1;
        if (l >= NL) l = NL - 1;
        int32_t pos = 0;
        unsigned long long todo = __ballot(l >= 0);
        while (todo) {   // one round per distinct length among the 64 rows (1-4 in practice)
            const int leader = __ffsll((long long)todo) - 1;
            const int lv = __shfl(l, leader, 64);
            const unsigned long long same = __ballot(l == lv);
            if (l == lv) pos = start[lv] + __popcll(same & ((1ull << lane) - 1ull));
            __syncthreads();
            if (lane == leader) start[lv] += __popcll(same);
            __syncthreads();
            todo &= ~same;
        }
        if (i < W) {
            rowof[base + pos] = (int32_t)(base + i);
            posof[base + i] = (int32_t)(base + pos);
        }
    }
}

// ---- step 2c: columns into the ELL slots.  One wavefront per slice.  The distinct columns of a row
// are a contiguous run in ucols; the slots want them transposed (slot-major, 64 rows side by side).
// The wave reads its 64 rows one after the other, each as ONE coalesced access, into an LDS tile, then
// writes slot by slot, every store a full 256-B line.  (The first version let every lane walk its own
// run: 64 lines touched per load, 8.4 GB fetched for 0.36 GB of columns, 1.2 ms at 148^3.)
__global__ void __launch_bounds__(256)
k_fill_cols(int32_t nslices, int64_t nloc, int64_t r0, int64_t r1, const int32_t *slot_ptr,
            const int32_t *rowof, const int32_t *rowlen, const int64_t *ptr, const int32_t *ucols,
            const int64_t *halo_rank, int32_t *cols, int32_t wtile) {
    extern __shared__ int32_t tile_all[];   // [4 waves][64 rows][wtile | 1] (odd stride: conflict-free column reads)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t slice = (int64_t)blockIdx.x * 4 + w;
    if (slice >= nslices) return;   // (no workgroup barrier below: the waves are independent)
    const int stride = wtile | 1;
    int32_t *tile = tile_all + (size_t)w * 64 * stride;
    const int64_t row = rowof[slice * 64 + lane];
    const bool live = row < nloc;   // rows >= nloc: padding of the last slice (zero values, column 0)
    const int rl = live ? rowlen[row] : 0;
    const int64_t p8 = live ? 8 * ptr[row] : 0;
    const int32_t k0 = slot_ptr[slice], k1 = slot_ptr[slice + 1];
    const int32_t own = live ? (int32_t)row : 0;   // a row shorter than its slice is padded with its own column
    // a slice wider than the tile (a high-valence node among its rows) goes through it in chunks of wtile slots
    for (int32_t c0 = 0; c0 < k1 - k0; c0 += wtile) {
        for (int r = 0; r < 64; r++) {
            const int rlr = __shfl(rl, r, 64);
            const int64_t pr = __shfl(p8, r, 64);
            const int hi = rlr < c0 + wtile ? rlr : c0 + wtile;
            for (int k = c0 + lane; k < hi; k += 64) {
                const int64_t g = ucols[pr + k];
                tile[r * stride + (k - c0)] = (g >= r0 && g < r1) ? (int32_t)(g - r0) : (int32_t)(nloc + halo_rank[g]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int32_t kend = k1 - k0 < c0 + wtile ? k1 - k0 : c0 + wtile;
        for (int32_t k = c0; k < kend; k++)
            cols[((int64_t)k0 + k) * 64 + lane] = k < rl ? tile[lane * stride + (k - c0)] : own;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// slice width = longest row of the slice; also accumulates block count and max width
// (4 slices per workgroup, one atomic pair per workgroup: 51 k same-address atomics were 1.2 ms)
__global__ void __launch_bounds__(256)
k_slice_width(int32_t nslices, const int32_t *rowlen, const int32_t *rowof, int32_t *width,
              unsigned long long *nblocks, int32_t *maxw) {
    __shared__ int sh_s[4], sh_m[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t slice = (int64_t)blockIdx.x * 4 + w;
    int v = slice < nslices ? rowlen[rowof[slice * 64 + lane]] : 0;
    int s = v;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        v = max(v, __shfl_xor(v, d, 64));
        s += __shfl_xor(s, d, 64);
    }
    if (lane == 0) {
        if (slice < nslices) width[slice] = v;
        sh_s[w] = s;
        sh_m[w] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(nblocks, (unsigned long long)(sh_s[0] + sh_s[1] + sh_s[2] + sh_s[3]));
        atomicMax(maxw, max(max(sh_m[0], sh_m[1]), max(sh_m[2], sh_m[3])));
    }
}

__global__ void k_count_fixed(int64_t n, const int32_t *red, unsigned long long *count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) c += red[i] == -1;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, (unsigned long long)c);
}

__global__ void k_i64_to_i32(const int64_t *in, int32_t *out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)in[i];
}

// halo_glob[rank] = g for every flagged g
__global__ void k_compact_flags(const int32_t *flag, const int64_t *rank, int32_t *out,
                                int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && flag[i]) out[rank[i]] = (int32_t)i;
}

// ---- step 3: numeric.  256 threads = 4 wavefronts assemble the block rows of 16 consecutive --
// positions of a slice (position -> row through rowof: SELL-C-sigma) ---------------------------
struct numeric_args {
    int64_t nloc, r0, r1, nhalo;
    const int64_t *ptr;
    const int32_t *list;
    const int32_t *crow;     // [n_elem*8] global block row of each element node (perm[conn])
    const double *xrow;      // [nb_glob*3] coordinates by global block row
    const int32_t *elem_mat;
    const uint8_t *elem_type;
    const double *mat_lamG;  // [n_mat*2] lambda, G
    const uint8_t *fixmask;  // by global block row
    const int32_t *halo_glob;
    const int64_t *halo_rank;
    const int32_t *rowlen;
    const int32_t *rowof;  // [nslices*64] position in the sliced layout -> local block row (k_window_sort)
    const int32_t *slot_ptr;
    const int32_t *cols;
    double *vals;
    long long *bad_elem;  // min element index with det J == 0, else LLONG_MAX
    int32_t wmax;         // LDS accumulators are sized for this slice width
    const int32_t *wide_slices;   // k_numeric_wide: the slices wider than wmax
};

// tuning constants of k_numeric (profiles/r04/k_numeric_ablations_2_and_tuning.txt; the timing-only ablations behind that
// file are the lab build's: lab/lab_hooks.patch)
#ifndef STAN_NUM_WAVES
#define STAN_NUM_WAVES 3   // workgroups of k_numeric per CU the register budget must allow (168 VGPRs)
#endif
#ifndef STAN_NUM_NT
#define STAN_NUM_NT 0      // non-temporal stores in k_numeric's write-out (lab)
#endif
#ifndef STAN_NUM_XS
#define STAN_NUM_XS 25     // doubles per incidence of the coordinate scratch (24 + padding against LDS bank conflicts)
#endif
#ifndef STAN_NUM_GS
#define STAN_NUM_GS 13     // doubles per (Gauss point, incidence) record (12 + padding)
#endif
#ifndef STAN_B_UNROLL
#define STAN_B_UNROLL 4    // Gauss points per trip of the block arithmetic's loop (2: 10.11 ms, 4: 9.98 ms)
#endif
#ifndef STAN_NUM_ROWS
#define STAN_NUM_ROWS 8    // block rows per workgroup of k_numeric (4 waves: 2 each); 16 = rounds 1-3
#endif
__global__ void __launch_bounds__(256, STAN_NUM_WAVES) k_numeric(numeric_args A) {
    extern __shared__ double lds[];
    // A workgroup assembles NR = 8 consecutive positions of a slice, two per wavefront (round 4; 16 until then: the
    // accumulators of 16 rows held a CU to two workgroups, and the kernel's time follows the waves in flight: 18.9 ms
    // with one workgroup per CU, 11.5 with two -- profiles/r04/k_numeric_ablations_*.txt).  The write-out then stores
    // 64-B half lines; the other half of a line follows from the neighbouring workgroup.
    // carve-up (all 8-byte aligned):
    //   acc   [NR][wmax][9]   double
    //   xs    [4 waves][8 inc][XS = 25]    double   8 nodes x 3 coordinates + 1 of padding
    //   gps   [4 waves][8 gp][8 inc][GS = 13] double  {J^-1 (9), c * grad N_a (3)} + 1 of padding
    // The paddings are LDS bank hygiene: the eight lanes of an incidence read ONE record, the wave eight records at a
    // time; with strides of 24 / 12 doubles the records of incidences s and s + 2 / s + 4 start in the same bank (4- / 2-way
    // conflicts on every read of the two read-heaviest phases); 25 / 13 spread the eight over distinct banks.
    //   colsl [NR][wmax] int32, cfix [NR][wmax] uint8 (stored as int32 for simplicity)
    constexpr int NR = STAN_NUM_ROWS, XS = STAN_NUM_XS, GS = STAN_NUM_GS;
    const int W = A.wmax;
    double *acc = lds;
    double *xs = acc + NR * W * 9;
    double *gps = xs + 4 * 8 * XS;
    int32_t *colsl = (int32_t *)(gps + 4 * 8 * 8 * GS);
    int32_t *cfix = colsl + NR * W;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // XCD-chunked mapping (round 4): with the identity mapping every XCD's L2 sees the coordinates and connectivity of
    // three whole breadth-first levels (4.8 MB at 148^3, L2: 4 MB); each XCD now assembles NUM_CH consecutive workgroups
    // (8 slices) at a time and its L2 holds their neighbourhood only.
#ifndef STAN_NUM_CH
#define STAN_NUM_CH 64
#endif
    const int64_t bid = STAN_NUM_CH > 0 ? xcd_chunked(blockIdx.x, gridDim.x, STAN_NUM_CH) : (int64_t)blockIdx.x;
    const int64_t slice = bid / (64 / NR);
    const int q = (int)(bid % (64 / NR));
    const int64_t row_base = slice * 64 + q * NR;   // first POSITION of this workgroup
    const int32_t k0 = A.slot_ptr[slice];
    const int sw = A.slot_ptr[slice + 1] - k0;  // this slice's width
    if (sw > W) return;   // wider than the LDS accumulators (a high-valence node): k_numeric_wide (workgroup-uniform)

    double *xsw = xs + w * (8 * XS);
    double *gpw = gps + w * (8 * 8 * GS);
    const int s = lane >> 3, b = lane & 7;
    // This lane's node b never changes: the signs of its natural coordinates are constants of the lane, and
    // dN_b/d(xi, eta, zeta) at a Gauss point is (sx/8) fy fz etc. with f = 1 +- gl picked at compile time per point
    // (hex8_dnl's own expression and association: same bits) -- 6 multiplications instead of ~20 instructions.
    const double sxb = hex8_sign(HEX8_SX, b), syb = hex8_sign(HEX8_SY, b), szb = hex8_sign(HEX8_SZ, b);
    const double sx8 = 0.125 * sxb, sy8 = 0.125 * syb, sz8 = 0.125 * szb;
    // The incidence -> connectivity -> coordinates chain is three dependent global loads: the chain of the NEXT row is
    // issued before the current row is computed (first 8 incidences; longer rows load the rest inline) -- one set of
    // registers, rotated (round 4: the four rows' chains up front cost 40 VGPRs that the block arithmetic needs).
    struct chain { int64_t row, p0; int deg, rl; int32_t en, colg, ty, mi; double x0, x1, x2; };
    auto load_chain = [&](int i) {
        chain c;
        c.row = i < NR / 4 ? (int64_t)A.rowof[row_base + w * (NR / 4) + i] : A.nloc;
        c.p0 = 0; c.deg = 0; c.rl = 0; c.en = 0; c.colg = 0; c.ty = STAN_HEX8_G2; c.mi = 0; c.x0 = c.x1 = c.x2 = 0.0;
        if (c.row < A.nloc) {
            c.p0 = A.ptr[c.row];
            c.deg = (int)(A.ptr[c.row + 1] - c.p0);
            c.rl = A.rowlen[c.row];
            if (s < c.deg) {
                c.en = A.list[c.p0 + s];
                const int32_t e = c.en >> 3;
                c.colg = A.crow[(int64_t)e * 8 + b];
                c.ty = A.elem_type[e];
                c.mi = A.elem_mat[e];
                c.x0 = A.xrow[3 * (int64_t)c.colg + 0];
                c.x1 = A.xrow[3 * (int64_t)c.colg + 1];
                c.x2 = A.xrow[3 * (int64_t)c.colg + 2];
            }
        }
        return c;
    };
    chain nxt = load_chain(0);   // (issued in front of the staging below: its five dependent loads overlap the staging's two)

    for (int i = tid; i < NR * W * 9; i += 256) acc[i] = 0.0;
    for (int i = tid; i < NR * sw; i += 256) {
        const int r16 = i % NR, k = i / NR;
        const int32_t lc = A.cols[((int64_t)k0 + k) * 64 + q * NR + r16];
        const int64_t g = lc < A.nloc ? A.r0 + lc : (int64_t)A.halo_glob[lc - A.nloc];
        colsl[r16 * W + k] = (int32_t)g;  // GLOBAL block column: ascending along the row
        cfix[r16 * W + k] = A.fixmask[g];
    }
    __syncthreads();

#pragma unroll 1
    for (int i = 0; i < NR / 4; i++) {
        const chain cur = nxt;
        nxt = load_chain(i + 1);
        const int r16 = w * (NR / 4) + i;
        const int64_t row = cur.row;
        if (row >= A.nloc) continue;  // wave-uniform
        const int64_t p0 = cur.p0;
        const int deg = cur.deg;
        const int rl = cur.rl;
        for (int c0 = 0; c0 < deg; c0 += 8) {
            const bool valid = c0 + s < deg;
            int32_t e = 0, a = 0, type = STAN_HEX8_G2, colg = 0;
            double lam = 0, G = 0;
            if (valid) {
                int32_t en, m;
                double x0, x1, x2;
                if (c0 == 0) {
                    en = cur.en; colg = cur.colg; type = cur.ty; m = cur.mi;
                    x0 = cur.x0; x1 = cur.x1; x2 = cur.x2;
                } else {
                    en = A.list[p0 + c0 + s];
                    colg = A.crow[(int64_t)(en >> 3) * 8 + b];
                    type = A.elem_type[en >> 3];
                    m = A.elem_mat[en >> 3];
                    x0 = A.xrow[3 * (int64_t)colg + 0];
                    x1 = A.xrow[3 * (int64_t)colg + 1];
                    x2 = A.xrow[3 * (int64_t)colg + 2];
                }
                e = en >> 3;
                a = en & 7;
                lam = A.mat_lamG[2 * m];
                G = A.mat_lamG[2 * m + 1];
                xsw[s * XS + b * 3 + 0] = x0;
                xsw[s * XS + b * 3 + 1] = x1;
                xsw[s * XS + b * 3 + 2] = x2;
            }
            // (all LDS traffic below is private to this wavefront: program order suffices,
            //  the fences only stop the compiler from reordering across the hand-off)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (valid) {
                // phase A: this lane = Gauss point b of incidence s: J^-1, c = det J w, and -- once for the eight lanes
                // that will need it -- c * grad N_a of the row's own node a
                double o[10], wa[3];
                const double det = hex8_gp_setup(xsw + s * XS, type, b, o);
                if (det == 0.0 && hex8_gauss_weight(type, b) != 0.0)
                    atomicMin(A.bad_elem, (long long)e);
                {
                    const double gl = hex8_gauss_loc(type);
                    hex8_wgrad(o, a, hex8_sign(HEX8_SX, b) * gl, hex8_sign(HEX8_SY, b) * gl, hex8_sign(HEX8_SZ, b) * gl, wa);
                }
#pragma unroll
                for (int j = 0; j < 9; j++) gpw[(b * 8 + s) * GS + j] = o[j];
#pragma unroll
                for (int j = 0; j < 3; j++) gpw[(b * 8 + s) * GS + 9 + j] = wa[j];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double kb[9];
            int pos = -1;
            if (valid) {
                // phase B: block (a, b) of element e in the M-form (hex8_device.h): per Gauss point J^-1 and c grad N_a
                // from LDS, grad N_b = J^-1 dnb[g] from the lane's constants, nine fused multiply-adds into M
                const double glt = hex8_gauss_loc(type);   // (HEX8_G1: 0 -- every point at the origin, c = 0 beyond the first)
                const double fxp = 1.0 + sxb * glt, fxm = 1.0 - sxb * glt, fyp = 1.0 + syb * glt, fym = 1.0 - syb * glt,
                             fzp = 1.0 + szb * glt, fzm = 1.0 - szb * glt;
                double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll STAN_B_UNROLL   // (fully unrolled the scheduler hoists all 96 LDS reads: 256 VGPRs and scratch)
                for (int g = 0; g < 8; g++) {
                    const double *q = gpw + (g * 8 + s) * GS;
                    const double fx = ((HEX8_SX >> g) & 1u) ? fxp : fxm, fy = ((HEX8_SY >> g) & 1u) ? fyp : fym,
                                 fz = ((HEX8_SZ >> g) & 1u) ? fzp : fzm;
                    const double d[3] = {sx8 * fy * fz, sy8 * fx * fz, sz8 * fx * fy};
                    hex8_m_accum(q, q + 9, d, M);
                }
                hex8_k_from_m(M, lam, G, kb);
                // phase C: slot of column colg in this row.  The row's columns ascend in GLOBAL
                // index (the symbolic phase sorted them), so a binary search over the global
                // indices staged in LDS finds it in log2(rl) steps, the same for every lane
                // (a linear scan made the wave wait for its slowest lane: 3.6 -> 0.9 ms at 148^3).
                const int32_t *cl = colsl + r16 * W;
                int lo = 0, hi = rl;  // invariant: cl[lo-1] < colg <= cl[hi]
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (cl[mid] < colg) lo = mid + 1; else hi = mid;
                }
                pos = (lo < rl && cl[lo] == colg) ? lo : -1;
            }
            // phase D: ordered accumulation, incidence by incidence (ascending element index)
            int isdup = 0;
            {
#pragma unroll
                for (int j = 0; j < 7; j++) {
                    const int pj = __shfl(pos, (lane & ~7) | j, 64);
                    if (j < b && pj == pos && pos >= 0) isdup = 1;
                }
            }
            const bool anydup = __ballot(isdup) != 0ull;
            {
                // One incidence after the other (ascending element index, then local node: the fixed order that makes K
                // bit-reproducible), each lane adding its block to its slot with LDS fp64 adds (ds_add_f64, no return
                // value): the eight lanes of one incidence hit eight different slots unless the element lists a node
                // twice, and the LDS executes a wave's instructions in program order, so the adds of successive rounds
                // to one slot land in that order.  Round 4: replaces the staged gather (stage 64 x 9 doubles, a slot map,
                // 72 reads per lane) -- the kernel is bound by the bytes it moves through the LDS pipe
                // (profiles/r04/k_numeric_ablations_*.txt), and this form moves a quarter of the gather's.
                double *ar = acc + r16 * W * 9;
#pragma unroll 1
                for (int s2 = 0; s2 < 8; s2++) {
                    if (s == s2 && pos >= 0 && !isdup) {
#pragma unroll
                        for (int j = 0; j < 9; j++)
                            __hip_atomic_fetch_add(ar + pos * 9 + j, kb[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                    if (anydup) {   // a degenerate element: the later listings of a node, in order
                        for (int b2 = 1; b2 < 8; b2++) {
                            if (s == s2 && b == b2 && pos >= 0 && isdup) {
#pragma unroll
                                for (int j = 0; j < 9; j++)
                                    __hip_atomic_fetch_add(ar + pos * 9 + j, kb[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            }
                        }
                    }
                    // the rounds must stay eight separate groups of DS instructions in this order (the summation order of a
                    // slot is the order of the rounds): no instruction moves across this point (per-thread
                    // semantics alone would let a compiler merge the predicated adds of different rounds)
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    __syncthreads();

    // write-out with the essential BCs applied: NR consecutive lanes = one 64-B half line (NR = 16: a full 128-B line)
    // a thread keeps its row (t % NR is loop-invariant) and walks (k, comp) by 256 / NR per step
    {
        const int r16 = tid % NR;
        const int64_t row = A.rowof[row_base + r16];
        const bool live = row < A.nloc;
        const int rfix = live ? A.fixmask[A.r0 + row] : 0;
        const int rlen = live ? A.rowlen[row] : 0;
        const int32_t grow = (int32_t)(A.r0 + row);
        constexpr int STEP = 256 / NR;     // (k, comp) pairs per step
        int kc = tid / NR;                 // = k * 9 + comp
        int k = kc / 9, comp = kc - 9 * k;
        double *out = A.vals + (int64_t)k0 * 9 * 64 + q * NR + r16;
        const int kc_end = sw * 9;
        for (; kc < kc_end; kc += STEP) {
            double v = 0.0;
            if (k < rlen) {
                const int m = comp >= 6 ? 2 : comp >= 3 ? 1 : 0, n = comp - 3 * m;
                const int cf = cfix[r16 * W + k];
                if (((rfix >> m) & 1) || ((cf >> n) & 1))
                    v = (colsl[r16 * W + k] == grow && m == n) ? 1.0 : 0.0;
                else
                    v = acc[(r16 * W + k) * 9 + comp];
            }
#if STAN_NUM_NT
            __builtin_nontemporal_store(v, out + (int64_t)kc * 64);   // K is written once and read by another kernel much later
#else
            out[(int64_t)kc * 64] = v;
#endif
            comp += STEP % 9;
            k += STEP / 9;
            if (comp >= 9) { comp -= 9; k += 1; }
        }
    }
}

// ---- step 3, wide slices.  A slice that holds a row of more than STAN_MAX_ROW_BLOCKS blocks (a high-valence node: the
// axis of a revolved mesh, the centre of a fan) does not fit the LDS accumulators above.  Slow path, one WAVEFRONT per
// row of such a slice, accumulating straight in K's values: slot k of the row belongs to lane k % 64, which zeroes it,
// adds every contribution to it (in the order of the fast path's ordered branch: ascending element, local node, then b)
// and applies the essential BCs -- one thread per address, so program order is all the ordering it needs.
__global__ void __launch_bounds__(64) k_numeric_wide(numeric_args A) {
    __shared__ double xsw[8 * 8 * 3];
    __shared__ double gpw[8 * 8 * 10];
    __shared__ double stage[64 * 9];
    __shared__ int32_t posl[64];
    const int lane = threadIdx.x;
    const int64_t slice = A.wide_slices[blockIdx.x >> 6];   // the slices k_numeric left out (k_list_above over the widths)
    const int r = blockIdx.x & 63;
    const int32_t k0 = A.slot_ptr[slice];
    const int sw = A.slot_ptr[slice + 1] - k0;
    const int64_t row = A.rowof[slice * 64 + r];
    double *vrow = A.vals + (int64_t)k0 * 9 * 64 + r;          // entry (k, comp) of this row: vrow[(k * 9 + comp) * 64]
    const int32_t *crow_cols = A.cols + (int64_t)k0 * 64 + r;  // local column of slot k: crow_cols[k * 64]
    for (int k = lane; k < sw; k += 64)
#pragma unroll
        for (int j = 0; j < 9; j++) vrow[(int64_t)(k * 9 + j) * 64] = 0.0;
    if (row >= A.nloc) return;   // padding row of the last slice
    const int64_t p0 = A.ptr[row];
    const int64_t deg = A.ptr[row + 1] - p0;
    const int rl = A.rowlen[row];
    auto gcol = [&](int k) -> int32_t {
        const int32_t lc = crow_cols[(int64_t)k * 64];
        return lc < A.nloc ? (int32_t)(A.r0 + lc) : A.halo_glob[lc - A.nloc];
    };
    const int s = lane >> 3, b = lane & 7;
    for (int64_t c0 = 0; c0 < deg; c0 += 8) {
        const bool valid = c0 + s < deg;
        int32_t e = 0, a = 0, type = STAN_HEX8_G2, colg = 0;
        double lam = 0, G = 0;
        if (valid) {
            const int32_t en = A.list[p0 + c0 + s];
            e = en >> 3;
            a = en & 7;
            colg = A.crow[(int64_t)e * 8 + b];
            type = A.elem_type[e];
            const int32_t m = A.elem_mat[e];
            lam = A.mat_lamG[2 * m];
            G = A.mat_lamG[2 * m + 1];
            xsw[(s * 8 + b) * 3 + 0] = A.xrow[3 * (int64_t)colg + 0];
            xsw[(s * 8 + b) * 3 + 1] = A.xrow[3 * (int64_t)colg + 1];
            xsw[(s * 8 + b) * 3 + 2] = A.xrow[3 * (int64_t)colg + 2];
        }
        __syncthreads();
        if (valid) {   // this lane = Gauss point b of incidence s
            double o[10];
            const double det = hex8_gp_setup(xsw + s * 24, type, b, o);
            if (det == 0.0 && hex8_gauss_weight(type, b) != 0.0) atomicMin(A.bad_elem, (long long)e);
#pragma unroll
            for (int j = 0; j < 10; j++) gpw[(b * 8 + s) * 10 + j] = o[j];
        }
        __syncthreads();
        int pos = -1;
        if (valid) {   // block (a, b) of element e and its slot in the row (columns ascend in global index)
            double kb[9];
            hex8_block_ab(gpw + s * 10, 8 * 10, type, a, b, lam, G, kb);
            int lo = 0, hi = rl;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (gcol(mid) < colg) lo = mid + 1; else hi = mid;
            }
            pos = (lo < rl && gcol(lo) == colg) ? lo : -1;
#pragma unroll
            for (int j = 0; j < 9; j++) stage[lane * 9 + j] = kb[j];
        }
        posl[lane] = pos;
        __syncthreads();
        for (int l = 0; l < 64; l++) {
            const int p = posl[l];
            if (p >= 0 && (p & 63) == lane) {
#pragma unroll
                for (int j = 0; j < 9; j++) vrow[(int64_t)(p * 9 + j) * 64] += stage[l * 9 + j];
            }
        }
        __syncthreads();
    }
    const int rfix = A.fixmask[A.r0 + row];
    const int32_t grow = (int32_t)(A.r0 + row);
    for (int k = lane; k < rl; k += 64) {
        const int32_t g = gcol(k);
        const int cf = A.fixmask[g];
        if (!rfix && !cf) continue;
#pragma unroll
        for (int m = 0; m < 3; m++)
#pragma unroll
            for (int n = 0; n < 3; n++)
                if (((rfix >> m) & 1) || ((cf >> n) & 1)) vrow[(int64_t)(k * 9 + 3 * m + n) * 64] = (g == grow && m == n) ? 1.0 : 0.0;
    }
}

// slice class: 1 if any row of the slice references a halo column (local index >= nloc)
__global__ void __launch_bounds__(64)
k_slice_class(int64_t nloc, const int32_t *rowlen, const int32_t *rowof, const int32_t *slot_ptr,
              const int32_t *cols, int32_t *is_bnd, int32_t *is_int) {
    const int lane = threadIdx.x;
    const int64_t slice = blockIdx.x;
    const int64_t row = rowof[slice * 64 + lane];
    const int32_t k0 = slot_ptr[slice];
    bool f = false;
    if (row < nloc)
        for (int k = 0; k < rowlen[row]; k++) f |= cols[((int64_t)k0 + k) * 64 + lane] >= nloc;
    const bool any = __ballot(f) != 0ull;
    if (lane == 0) { is_bnd[slice] = any ? 1 : 0; is_int[slice] = any ? 0 : 1; }
}

// ---- per-row rank mask (which ranks need this owned row's x) --------------------------------
__global__ void k_row_rankflag(int64_t nloc, const int32_t *rowlen, const int32_t *posof,
                               const int32_t *slot_ptr, const int32_t *cols, const int32_t *halo_glob,
                               int64_t q0, int64_t q1, int32_t *flag) {
    int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nloc) return;
    const int64_t pos = posof[row];
    const int64_t slice = pos >> 6;
    const int lane = (int)(pos & 63);
    const int32_t k0 = slot_ptr[slice];
    int f = 0;
    for (int k = 0; k < rowlen[row]; k++) {
        const int32_t lc = cols[((int64_t)k0 + k) * 64 + lane];
        if (lc >= nloc) {
            const int64_t g = halo_glob[lc - nloc];
            if (g >= q0 && g < q1) f = 1;
        }
    }
    flag[row] = f;
}

// ---- debug / parity: K_e of whole elements, one wavefront per element -------------------------
__global__ void __launch_bounds__(64)
k_ke_batch(int64_t n, const double *xyz8, double lam, double G, const uint8_t *type,
           double *out, long long *bad_elem) {
    __shared__ double xs[24];
    __shared__ double gp[8 * 10];
    __shared__ double ke[576];  // the 24x24 K_e staged in LDS, then stored as whole lines
    const int lane = threadIdx.x;
    const int64_t e = blockIdx.x;
    if (e >= n) return;
    const int t = type[e];
    if (lane < 24) xs[lane] = xyz8[e * 24 + lane];
    __syncthreads();
    if (lane < 8) {
        double o[10];
        const double det = hex8_gp_setup(xs, t, lane, o);
        if (det == 0.0 && hex8_gauss_weight(t, lane) != 0.0) atomicMin(bad_elem, (long long)e);
        for (int j = 0; j < 10; j++) gp[lane * 10 + j] = o[j];
    }
    __syncthreads();
    const int a = lane >> 3, b = lane & 7;
    double kb[9];
    hex8_block_ab(gp, 10, t, a, b, lam, G, kb);
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int nn = 0; nn < 3; nn++) ke[(3 * a + m) * 24 + 3 * b + nn] = kb[3 * m + nn];
    __syncthreads();
    for (int i = lane; i < 576; i += 64) out[e * 576 + i] = ke[i];
}

inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

struct tmp_free {  // releases device temporaries on every exit path (to the context's pool)
    stan_ctx *ctx = nullptr;
    std::vector<void *> p;
    ~tmp_free() {
        for (void *q : p) stan_dfree(ctx, q);
    }
    template <typename T>
    void own(T *q) { p.push_back((void *)q); }
};

}  // namespace

int stan_ke_batch_device(stan_ctx *ctx, int64_t n, const double *d_xyz8, double E, double nu,
                         const uint8_t *d_type, double *d_out) {
    double lam, G;
    stan_lame(E, nu, &lam, &G);
    long long init = 0x7fffffffffffffffLL;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_status + SS_BAD_ELEM, &init, 8, hipMemcpyHostToDevice, ctx->stream));
    if (n > 0)
        hipLaunchKernelGGL(k_ke_batch, dim3((unsigned)n), dim3(64), 0, ctx->stream, n, d_xyz8, lam,
                           G, d_type, d_out, (long long *)(ctx->d_status + SS_BAD_ELEM));
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_status + SS_BAD_ELEM, ctx->d_status + SS_BAD_ELEM, 8, hipMemcpyDeviceToHost,
                               ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->h_status[SS_BAD_ELEM] != init) {
        ctx->bad_elem = ctx->h_status[SS_BAD_ELEM];
        ctx->err = "det J == 0 in element " + std::to_string(ctx->bad_elem) +
                   " (MatrixST.Inverse would throw)";
        return STAN_E_DETJ;
    }
    return STAN_OK;
}

int stan_assemble_device(stan_ctx *ctx, int64_t n_nodes, const double *d_xyz,
                         const int32_t *d_node_dof, int64_t n_elem, const int32_t *d_conn,
                         const int32_t *d_elem_mat, const uint8_t *d_elem_type, int32_t n_mat,
                         const double *mat_E_nu, int64_t n_dof, const int32_t *d_red,
                         stan_matrix **outK) {
    *outK = nullptr;
    if (n_nodes <= 0 || n_elem < 0 || n_dof != 3 * n_nodes || n_mat <= 0) {
        ctx->err = "assemble: need n_nodes > 0, n_dof == 3*n_nodes, n_mat > 0";
        return STAN_E_ARG;
    }
    if (n_elem * 8 >= (int64_t)1 << 31) {
        ctx->err = "assemble: n_elem*8 must fit int32";
        return STAN_E_ARG;
    }
    hipStream_t st = ctx->stream;
    event_bag events;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    if (ctx->profiling) {
        ev0 = events.make(); ev1 = events.make(); ev2 = events.make();
        hipEventRecord(ev0, st);
    }
    tmp_free tmp;
    tmp.ctx = ctx;
    stan_matrix *K = new stan_matrix();
    K->ctx = ctx;
    ctx->matrices.push_back(K);
    struct guard {
        stan_matrix *k; bool ok = false;
        ~guard() { if (!ok) stan_hip_matrix_free(k); }
    } g{K};
    const int64_t nb = n_dof / 3;
    K->n_dof = n_dof;
    K->nb_glob = nb;
    K->n_elem_scanned = n_elem;
    // contiguous block-row partition, cut on slice boundaries
    K->row_starts.resize(ctx->nranks + 1);
    for (int r = 0; r <= ctx->nranks; r++) K->row_starts[r] = stan_row_start(nb, ctx->nranks, r);
    const int64_t r0 = K->r0 = K->row_starts[ctx->rank], r1 = K->r1 = K->row_starts[ctx->rank + 1];
    const int64_t nloc = K->nloc = r1 - r0;
    K->nslices = (int32_t)((nloc + 63) / 64);
    const int64_t nrows_pad = (int64_t)K->nslices * 64;

    int64_t *d_status = ctx->d_status;
    HIPCHK(ctx, hipMemsetAsync(d_status, 0, 16 * 8, st));
    {
        long long init = 0x7fffffffffffffffLL;
        HIPCHK(ctx, hipMemcpyAsync(d_status + SS_BAD_ELEM, &init, 8, hipMemcpyHostToDevice, st));
    }

    // materials -> (lambda, G)
    std::vector<double> lamG(2 * (size_t)n_mat);
    for (int m = 0; m < n_mat; m++) stan_lame(mat_E_nu[2 * m], mat_E_nu[2 * m + 1], &lamG[2 * m], &lamG[2 * m + 1]);
    double *d_lamG; STANCHK(stan_dmalloc(ctx, &d_lamG, lamG.size())); tmp.own(d_lamG);
    HIPCHK(ctx, hipMemcpyAsync(d_lamG, lamG.data(), lamG.size() * 8, hipMemcpyHostToDevice, st));

    int32_t *d_perm; STANCHK(stan_dmalloc(ctx, &d_perm, (size_t)n_nodes)); tmp.own(d_perm);
    STANCHK(stan_dmalloc(ctx, &K->d_fixmask, (size_t)nb));
    STANCHK(stan_dmalloc(ctx, &K->d_red, (size_t)n_dof));
    HIPCHK(ctx, hipMemsetAsync(K->d_fixmask, 0, (size_t)nb, st));
    HIPCHK(ctx, hipMemcpyAsync(K->d_red, d_red, (size_t)n_dof * 4, hipMemcpyDeviceToDevice, st));
    double *d_xrow; STANCHK(stan_dmalloc(ctx, &d_xrow, (size_t)nb * 3)); tmp.own(d_xrow);
    int32_t *d_crow; STANCHK(stan_dmalloc(ctx, &d_crow, (size_t)(n_elem > 0 ? n_elem * 8 : 1))); tmp.own(d_crow);
    hipLaunchKernelGGL(k_perm, dim3(nblk(n_nodes, 256)), dim3(256), 0, st, n_nodes, nb, d_node_dof,
                       d_red, d_xyz, d_perm, K->d_fixmask, d_xrow, d_status);

    // incidence lists of owned rows
    int32_t *d_cnt; STANCHK(stan_dmalloc(ctx, &d_cnt, (size_t)nrows_pad + 1)); tmp.own(d_cnt);
    int32_t *d_big_rows = nullptr;   // rows with more than STAN_MAX_INCIDENT incidences (listed before d_cnt becomes the fill cursor)
    int64_t *d_ptr; STANCHK(stan_dmalloc(ctx, &d_ptr, (size_t)nrows_pad + 2)); tmp.own(d_ptr);
    HIPCHK(ctx, hipMemsetAsync(d_cnt, 0, ((size_t)nrows_pad + 1) * 4, st));
    if (n_elem > 0)
        hipLaunchKernelGGL(k_count_incident, dim3(nblk(n_elem * 8, 256)), dim3(256), 0, st, n_elem,
                           n_nodes, d_conn, d_perm, r0, r1, d_cnt, d_crow, d_status);
    constexpr int BIG_LDS_INTS = 32768;   // 128 KB of the CU's 160 KB for k_symbolic_big's two sort buffers (up to 3640 incidences)
    if (nrows_pad > 0)
        hipLaunchKernelGGL(k_max_incident, dim3(nblk(nrows_pad, 256)), dim3(256), 0, st, nrows_pad, d_cnt, BIG_LDS_INTS, d_status);
    STANCHK(stan_scan_exclusive(ctx, d_cnt, d_ptr, nrows_pad));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_status + SS_ERRBITS, d_status + SS_ERRBITS, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_status + SS_H_NINC, d_ptr + nrows_pad, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_status + SS_MAXDEG, d_status + SS_MAXDEG, 24, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (ctx->h_status[SS_ERRBITS] & ERR_DOF_LAYOUT) {
        ctx->err = "assemble: Node.DOF is not {3i,3i+1,3i+2} with 3i < n_dof (Node.cs:218-223)";
        return STAN_E_DOF_LAYOUT;
    }
    if (ctx->h_status[SS_ERRBITS] & ERR_CONN_RANGE) {
        ctx->err = "assemble: connectivity references a node index outside [0,n_nodes)";
        return STAN_E_ARG;
    }
    const int64_t n_inc = ctx->h_status[SS_H_NINC];
    const int64_t n_big = ctx->h_status[SS_NBIG];
    if (n_big > 0) {
        STANCHK(stan_dmalloc(ctx, &d_big_rows, (size_t)n_big)); tmp.own(d_big_rows);
        HIPCHK(ctx, hipMemsetAsync(d_status + SS_COUNTER, 0, 8, st));
        hipLaunchKernelGGL(k_list_above, dim3(nblk(nrows_pad, 256)), dim3(256), 0, st, nrows_pad, d_cnt, STAN_MAX_INCIDENT, d_big_rows,
                           (unsigned long long *)(d_status + SS_COUNTER));
    }
    int32_t *d_list; STANCHK(stan_dmalloc(ctx, &d_list, (size_t)n_inc)); tmp.own(d_list);
    HIPCHK(ctx, hipMemsetAsync(d_cnt, 0, ((size_t)nrows_pad + 1) * 4, st));
    if (n_elem > 0)
        hipLaunchKernelGGL(k_fill_incident, dim3(nblk(n_elem * 8, 256)), dim3(256), 0, st, n_elem,
                           d_crow, r0, r1, d_ptr, d_cnt, d_list);

    // symbolic count
    STANCHK(stan_dmalloc(ctx, &K->d_rowlen, (size_t)nrows_pad));
    int32_t *d_refflag = nullptr; int64_t *d_halo_rank = nullptr;
    if (ctx->nranks > 1) {
        STANCHK(stan_dmalloc(ctx, &d_refflag, (size_t)nb)); tmp.own(d_refflag);
        STANCHK(stan_dmalloc(ctx, &d_halo_rank, (size_t)nb + 1)); tmp.own(d_halo_rank);
        HIPCHK(ctx, hipMemsetAsync(d_refflag, 0, (size_t)nb * 4, st));
    }
    // distinct columns of row i land in d_ucols[8 * d_ptr[i] ...] (8 candidates per incidence)
    int32_t *d_ucols; STANCHK(stan_dmalloc(ctx, &d_ucols, (size_t)(n_inc > 0 ? 8 * n_inc : 1))); tmp.own(d_ucols);
    if (nrows_pad > 0)
        hipLaunchKernelGGL(k_symbolic, dim3((unsigned)nrows_pad), dim3(64), 0, st, nloc, r0, r1, d_ptr,
                           d_list, d_crow, K->d_rowlen, d_refflag, d_ucols, d_status);
    if (n_big > 0) {   // high-valence nodes: the slow symbolic path, one workgroup per listed row
        int64_t PD = 64, PC = 64;
        while (PD < ctx->h_status[SS_MAXDEG]) PD <<= 1;
        while (PC < 8 * ctx->h_status[SS_MAXDEG]) PC <<= 1;
        if (PC >= (int64_t)1 << 30) { ctx->err = "assemble: a node with more than 2^26 incident elements"; return STAN_E_VALENCE; }
        const int64_t lds_ints = PD + PC < BIG_LDS_INTS ? PD + PC : BIG_LDS_INTS;
        int32_t *d_scratch = nullptr;
        if (ctx->h_status[SS_NGIANT] > 0) {
            STANCHK(stan_dmalloc(ctx, &d_scratch, (size_t)(ctx->h_status[SS_NGIANT] * (PD + PC))));
            tmp.own(d_scratch);
        }
        HIPCHK(ctx, hipMemsetAsync(d_status + SS_COUNTER, 0, 8, st));   // scratch tickets
        if (lds_ints * 4 > 64 * 1024)
            HIPCHK(ctx, hipFuncSetAttribute((const void *)k_symbolic_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_ints * 4)));
        hipLaunchKernelGGL(k_symbolic_big, dim3((unsigned)n_big), dim3(256), (size_t)lds_ints * 4, st, d_big_rows, r0, r1, d_ptr, d_list,
                           d_crow, K->d_rowlen, d_refflag, d_ucols, (int)lds_ints, d_scratch, PD + PC,
                           (unsigned long long *)(d_status + SS_COUNTER));
    }
    // SELL-C-sigma: positions of the rows inside the sliced layout
    K->sigma = ctx->sell_sigma < 1 ? 1 : ctx->sell_sigma > 32 ? 32 : ctx->sell_sigma;
    STANCHK(stan_dmalloc(ctx, &K->d_rowof, (size_t)(nrows_pad > 0 ? nrows_pad : 1)));
    STANCHK(stan_dmalloc(ctx, &K->d_posof, (size_t)(nrows_pad > 0 ? nrows_pad : 1)));
    if (nrows_pad > 0)
        hipLaunchKernelGGL(k_window_sort, dim3(nblk(K->nslices, K->sigma)), dim3(64), 0, st, nrows_pad, K->sigma,
                           K->d_rowlen, K->d_rowof, K->d_posof);
    // halo numbering
    K->nhalo = 0;
    if (ctx->nranks > 1) {
        STANCHK(stan_scan_exclusive(ctx, d_refflag, d_halo_rank, nb));
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_status + SS_H_NHALO, d_halo_rank + nb, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        K->nhalo = ctx->h_status[SS_H_NHALO];
        STANCHK(stan_dmalloc(ctx, &K->d_halo_glob, (size_t)K->nhalo));
        hipLaunchKernelGGL(k_compact_flags, dim3(nblk(nb, 256)), dim3(256), 0, st, d_refflag,
                           d_halo_rank, K->d_halo_glob, nb);
    }
    // slice widths -> slot pointers
    int32_t *d_width; STANCHK(stan_dmalloc(ctx, &d_width, (size_t)K->nslices + 1)); tmp.own(d_width);
    int64_t *d_sp64; STANCHK(stan_dmalloc(ctx, &d_sp64, (size_t)K->nslices + 2)); tmp.own(d_sp64);
    HIPCHK(ctx, hipMemsetAsync(d_status + SS_WIDTH_SUM, 0, 16, st));
    if (K->nslices > 0)
        hipLaunchKernelGGL(k_slice_width, dim3(nblk(K->nslices, 4)), dim3(256), 0, st, K->nslices, K->d_rowlen,
                           K->d_rowof, d_width, (unsigned long long *)(d_status + SS_WIDTH_SUM), (int32_t *)(d_status + SS_WIDTH_MAX));
    STANCHK(stan_scan_exclusive(ctx, d_width, d_sp64, K->nslices));
    STANCHK(stan_dmalloc(ctx, &K->d_slot_ptr, (size_t)K->nslices + 1));
    hipLaunchKernelGGL(k_i64_to_i32, dim3(nblk(K->nslices + 1, 256)), dim3(256), 0, st, d_sp64,
                       K->d_slot_ptr, (int64_t)K->nslices + 1);
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_status + SS_ERRBITS, d_status + SS_ERRBITS, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_status + SS_H_NSLOTS, d_sp64 + K->nslices, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_status + SS_H_NBLOCKS, d_status + SS_WIDTH_SUM, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    K->nslots = ctx->h_status[SS_H_NSLOTS];
    K->nblocks = ctx->h_status[SS_H_NBLOCKS];
    K->max_row_blocks = (int32_t)(ctx->h_status[SS_H_MAXROW] & 0xffffffff);
    if (K->nslots * 64 >= (int64_t)1 << 31) {
        // cols index fits, but keep slot arithmetic in int32 honest
        ctx->err = "assemble: more than 2^31 ELL entries on one rank";
        return STAN_E_ARG;
    }
    STANCHK(stan_dmalloc(ctx, &K->d_cols, (size_t)K->nslots * 64));
    if (K->nslices > 0) {
        int32_t wmax = K->max_row_blocks > 0 ? K->max_row_blocks : 1;
        if (wmax > STAN_MAX_ROW_BLOCKS) wmax = STAN_MAX_ROW_BLOCKS;         // wider slices pass through the tile in chunks
        const size_t lds = (size_t)4 * 64 * (wmax | 1) * sizeof(int32_t);   // <= 4 * 64 * 97 * 4 = 99 KB
        if (lds > 64 * 1024)
            HIPCHK(ctx, hipFuncSetAttribute((const void *)k_fill_cols, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_fill_cols, dim3(nblk(K->nslices, 4)), dim3(256), lds, st, K->nslices, nloc, r0, r1,
                           K->d_slot_ptr, K->d_rowof, K->d_rowlen, d_ptr, d_ucols, (const int64_t *)d_halo_rank,
                           K->d_cols, wmax);
    }
    // (after the columns are in place: an allocation by trial times the SpMV itself, which needs the columns)
    STANCHK(stan_dmalloc_streamed(ctx, (void **)&K->d_vals, (size_t)K->nslots * 9 * 64 * 8,
                                  [&](const void *q, float *ms, bool self) {
                                      return stan_spmv_probe(ctx, K, q, (size_t)K->nslots * 9 * 64 * 8, STAN_PREC_FP64, ms, self);
                                  }));
    if (ctx->profiling) hipEventRecord(ev1, st);

    // numeric
    if (ctx->assembly_mode == 1) {
        if (ctx->nranks > 1) { ctx->err = "assembly mode 1 (colour scatter) is single-rank only"; return STAN_E_UNSUPPORTED; }
        STANCHK(stan_assemble_colour_scatter(ctx, K, n_elem, d_conn, d_perm, d_xyz, d_elem_mat, d_elem_type,
                                             d_lamG, d_ptr, d_list, (long long *)(d_status + SS_BAD_ELEM)));
    } else {
        ctx->prof_colours = 0;
        numeric_args A;
        A.nloc = nloc; A.r0 = r0; A.r1 = r1; A.nhalo = K->nhalo;
        A.ptr = d_ptr; A.list = d_list; A.crow = d_crow; A.xrow = d_xrow;
        A.elem_mat = d_elem_mat; A.elem_type = d_elem_type; A.mat_lamG = d_lamG;
        A.fixmask = K->d_fixmask; A.halo_glob = K->d_halo_glob; A.halo_rank = d_halo_rank;
        A.rowlen = K->d_rowlen; A.rowof = K->d_rowof; A.slot_ptr = K->d_slot_ptr; A.cols = K->d_cols; A.vals = K->d_vals;
        A.bad_elem = (long long *)(d_status + SS_BAD_ELEM);
        A.wide_slices = nullptr;
        A.wmax = K->max_row_blocks > 0 ? K->max_row_blocks : 1;
        const bool wide = A.wmax > STAN_MAX_ROW_BLOCKS;   // some slice holds a high-valence row: it goes to k_numeric_wide
        if (wide) A.wmax = STAN_MAX_ROW_BLOCKS;
        const size_t lds = (size_t)STAN_NUM_ROWS * A.wmax * 9 * 8 + (size_t)4 * 8 * STAN_NUM_XS * 8 +
                           (size_t)4 * 8 * 8 * STAN_NUM_GS * 8 + (size_t)2 * STAN_NUM_ROWS * A.wmax * 4;
        size_t lds_launch = lds;
        if (lds_launch > 64 * 1024)
            HIPCHK(ctx, hipFuncSetAttribute((const void *)k_numeric,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_launch));
        if (K->nslices > 0)
            hipLaunchKernelGGL(k_numeric, dim3((unsigned)K->nslices * (64 / STAN_NUM_ROWS)), dim3(256), lds_launch, st, A);
        if (wide) {   // the slices k_numeric skipped, listed from their widths (a handful: one launch of 64 waves per slice)
            int32_t *d_wide; STANCHK(stan_dmalloc(ctx, &d_wide, (size_t)K->nslices)); tmp.own(d_wide);
            HIPCHK(ctx, hipMemsetAsync(d_status + SS_COUNTER, 0, 8, st));
            hipLaunchKernelGGL(k_list_above, dim3(nblk(K->nslices, 256)), dim3(256), 0, st, (int64_t)K->nslices, d_width, STAN_MAX_ROW_BLOCKS, d_wide,
                               (unsigned long long *)(d_status + SS_COUNTER));
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_status + SS_COUNTER, d_status + SS_COUNTER, 8, hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            const int64_t n_wide = ctx->h_status[SS_COUNTER];
            A.wide_slices = d_wide;
            if (n_wide * 64 >= (int64_t)1 << 31) { ctx->err = "assemble: too many wide slices for one launch"; return STAN_E_ARG; }
            if (n_wide > 0) hipLaunchKernelGGL(k_numeric_wide, dim3((unsigned)(n_wide * 64)), dim3(64), 0, st, A);
        }
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemsetAsync(d_status + SS_AUX, 0, 8, st));
    hipLaunchKernelGGL(k_count_fixed, dim3(nblk(n_dof, 256) > 2048 ? 2048 : nblk(n_dof, 256)), dim3(256), 0, st, n_dof, d_red,
                       (unsigned long long *)(d_status + SS_AUX));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_status + SS_BAD_ELEM, d_status + SS_BAD_ELEM, 16, hipMemcpyDeviceToHost, st));

    // interior / boundary slice lists for the overlapped SpMV (also built for a 1-rank
    // communicator, where every slice is interior, so that the two-stream path can be tested)
    if ((ctx->nranks > 1 || ctx->comm) && K->nslices > 0) {
        int32_t *d_fb, *d_fi; int64_t *d_sb, *d_si;
        STANCHK(stan_dmalloc(ctx, &d_fb, (size_t)K->nslices + 1)); tmp.own(d_fb);
        STANCHK(stan_dmalloc(ctx, &d_fi, (size_t)K->nslices + 1)); tmp.own(d_fi);
        STANCHK(stan_dmalloc(ctx, &d_sb, (size_t)K->nslices + 2)); tmp.own(d_sb);
        STANCHK(stan_dmalloc(ctx, &d_si, (size_t)K->nslices + 2)); tmp.own(d_si);
        hipLaunchKernelGGL(k_slice_class, dim3((unsigned)K->nslices), dim3(64), 0, st, nloc,
                           K->d_rowlen, K->d_rowof, K->d_slot_ptr, K->d_cols, d_fb, d_fi);
        STANCHK(stan_scan_exclusive(ctx, d_fb, d_sb, K->nslices));
        STANCHK(stan_scan_exclusive(ctx, d_fi, d_si, K->nslices));
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_status + SS_H_COUNT_A, d_sb + K->nslices, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_status + SS_H_COUNT_B, d_si + K->nslices, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        K->n_sl_bnd = (int32_t)ctx->h_status[SS_H_COUNT_A];
        K->n_sl_int = (int32_t)ctx->h_status[SS_H_COUNT_B];
        STANCHK(stan_dmalloc(ctx, &K->d_sl_bnd, (size_t)K->n_sl_bnd));
        STANCHK(stan_dmalloc(ctx, &K->d_sl_int, (size_t)K->n_sl_int));
        hipLaunchKernelGGL(k_compact_flags, dim3(nblk(K->nslices, 256)), dim3(256), 0, st, d_fb, d_sb,
                           K->d_sl_bnd, (int64_t)K->nslices);
        hipLaunchKernelGGL(k_compact_flags, dim3(nblk(K->nslices, 256)), dim3(256), 0, st, d_fi, d_si,
                           K->d_sl_int, (int64_t)K->nslices);
    }
    // halo exchange plan
    if (ctx->nranks > 1) {
        std::vector<int32_t> hg((size_t)K->nhalo);
        HIPCHK(ctx, hipMemcpyAsync(hg.data(), K->d_halo_glob, hg.size() * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        int32_t *d_flag; STANCHK(stan_dmalloc(ctx, &d_flag, (size_t)nloc + 1)); tmp.own(d_flag);
        int64_t *d_rk; STANCHK(stan_dmalloc(ctx, &d_rk, (size_t)nloc + 2)); tmp.own(d_rk);
        std::vector<int32_t *> lists;
        std::vector<int64_t> counts;
        K->recv_off.push_back(0);
        K->send_off.push_back(0);
        for (int q = 0; q < ctx->nranks; q++) {
            if (q == ctx->rank) continue;
            const int64_t q0 = K->row_starts[q], q1 = K->row_starts[q + 1];
            // halo columns owned by q: contiguous in the (ascending) halo list
            int64_t lo = std::lower_bound(hg.begin(), hg.end(), (int32_t)q0) - hg.begin();
            int64_t hi = std::lower_bound(hg.begin(), hg.end(), (int32_t)q1) - hg.begin();
            if (q1 > 0x7fffffff) hi = (int64_t)hg.size();
            if (hi == lo) continue;  // structural symmetry: no recv <=> no send
            hipLaunchKernelGGL(k_row_rankflag, dim3(nblk(nloc, 256)), dim3(256), 0, st, nloc,
                               K->d_rowlen, K->d_posof, K->d_slot_ptr, K->d_cols, K->d_halo_glob, q0, q1, d_flag);
            STANCHK(stan_scan_exclusive(ctx, d_flag, d_rk, nloc));
            HIPCHK(ctx, hipMemcpyAsync(ctx->h_status + SS_H_COUNT_A, d_rk + nloc, 8, hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            const int64_t ns = ctx->h_status[SS_H_COUNT_A];
            int32_t *d_l; STANCHK(stan_dmalloc(ctx, &d_l, (size_t)ns)); tmp.own(d_l);
            hipLaunchKernelGGL(k_compact_flags, dim3(nblk(nloc, 256)), dim3(256), 0, st, d_flag, d_rk,
                               d_l, nloc);
            lists.push_back(d_l);
            counts.push_back(ns);
            K->nbr.push_back(q);
            K->recv_off.push_back(hi);
            K->send_off.push_back(K->send_off.back() + ns);
            // recv segment of q starts at lo: halo list is grouped by owner in rank order
            if ((int64_t)K->recv_off[K->recv_off.size() - 2] != lo) {
                ctx->err = "assemble: internal halo plan inconsistency";
                return STAN_E_COMM;
            }
        }
        const int64_t stot = K->send_off.back();
        STANCHK(stan_dmalloc(ctx, &K->d_send_rows, (size_t)stot));
        STANCHK(stan_dmalloc(ctx, &K->d_sendbuf, (size_t)stot * 3));
        for (size_t i = 0; i < lists.size(); i++)
            HIPCHK(ctx, hipMemcpyAsync(K->d_send_rows + K->send_off[i], lists[i], (size_t)counts[i] * 4,
                                       hipMemcpyDeviceToDevice, st));
    }
    if (ctx->profiling) hipEventRecord(ev2, st);
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (ctx->h_status[SS_BAD_ELEM] != 0x7fffffffffffffffLL) {
        ctx->bad_elem = ctx->h_status[SS_BAD_ELEM];
        ctx->err = "det J == 0 in element " + std::to_string(ctx->bad_elem) +
                   " (MatrixST.Inverse would throw, MatrixST.cs:315-318)";
        return STAN_E_DETJ;
    }
    K->n_red = n_dof - ctx->h_status[SS_AUX];
    if (ctx->profiling) {
        float a = 0, b = 0;
        hipEventElapsedTime(&a, ev0, ev1);
        hipEventElapsedTime(&b, ev1, ev2);
        ctx->prof.symbolic_ms = a;
        ctx->prof.numeric_ms = b;
        ctx->prof.assemble_ms = a + b;
    }
    g.ok = true;
    *outK = K;
    return STAN_OK;
}
