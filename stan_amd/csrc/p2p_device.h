// p2p_device.h -- device side of the peer-to-peer exchanges (internal.h: stan_p2p_dev).
// The reference has no distributed path (Solver.cs:156,162 are the only call sites of the hot
// path, one process): this is new design, SURVEY.md sections 5 and 8e ("avoid ring all-reduce for
// 16-byte payloads -- single-kernel one-shot", "or peer-mapped buffers").
//
// Memory model used here (system scope throughout; nothing relies on one GPU's L2):
//   producer   sc0 sc1 (write-through) stores of the payload -> release fence at system scope
//              (s_waitcnt vmcnt(0) behind it: the stores are acknowledged by their destination)
//              -> system-scope atomic add to the consumer's arrival counter
//   consumer   the STREAM waits for the counter (a one-wave polling kernel; optionally
//              hipStreamWaitValue64); the kernel behind the wait starts with the usual acquire and reads the
//              mailbox with sc0 sc1 loads (fine-grained memory: never served from a stale L2 line)
#pragma once
#include "internal.h"

__device__ __forceinline__ void st_sys(double *p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double ld_sys(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Where a reduction's result goes when the sharded CG exchanges peer to peer.
struct p2p_out {
    const stan_p2p_dev *pp;   // nullptr: not peer to peer (the scalar is written locally)
    int32_t slot;             // mailbox slot / counter of this exchange (0 .. RING-1)
    int32_t j0;               // first mailbox column of the values (a mailbox entry holds 4 doubles per rank)
    int32_t signal;           // 1: count this rank into every rank's arrival counter after storing
};

// One partial result r[0..NV) of THIS rank into every rank's mailbox (+ arrival count).  Called by
// all threads of one block with r valid in thread 0; `sh` = NV doubles of LDS.
template <int NV>
__device__ __forceinline__ void p2p_publish(const p2p_out &o, const double r[NV], double *sh) {
    const stan_p2p_dev *pp = o.pp;
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int j = 0; j < NV; j++) sh[j] = r[j];
    __syncthreads();
    const int q = threadIdx.x;   // thread q serves rank q: its stores, its fence, its count
    if (q < pp->n) {
        double *dst = pp->mbox[q] + ((int64_t)o.slot * pp->n + pp->me) * 4 + o.j0;
#pragma unroll
        for (int j = 0; j < NV; j++) st_sys(dst + j, sh[j]);
        if (o.signal) {
            __atomic_thread_fence(__ATOMIC_RELEASE);   // system scope: the stores above have arrived
            __hip_atomic_fetch_add(pp->sig_red[q][o.slot], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Where a consuming kernel finds the sums of a reduction: the local scalar (one rank, or an RCCL
// all-reduce wrote it) or the partials of all ranks in its mailbox slot, added in RANK ORDER --
// every rank gets the same bits, and the same bits as a rank-ordered all-reduce.
struct red_src {
    const double *mb;   // own mailbox + slot * n * 4; nullptr: read the scalar
    int32_t n;
    // STAN_P2P_WAIT_MODE=2: no wait was enqueued in front of this kernel -- it waits itself ("single-kernel
    // one-shot"): thread 0 of every block polls the arrival counter until it has reached `want`.  Safe for the
    // mailbox only: its entries are read with system-scope loads from fine-grained memory, whereas a halo region
    // written by a peer becomes visible to plain loads at a kernel boundary only (that wait stays a launch).
    const unsigned long long *ctr;
    unsigned long long want;
};
// NV values for the whole block; `sh` = NV doubles of LDS; call from ALL threads (one barrier).
template <int NV>
__device__ __forceinline__ void red_get(const double *scalars, const red_src &rs, double out[NV], double *sh) {
    if (!rs.mb) {
#pragma unroll
        for (int j = 0; j < NV; j++) out[j] = scalars[j];
        return;
    }
    if (rs.ctr) {   // block-uniform
        if (threadIdx.x == 0)
            while (__hip_atomic_load(rs.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < rs.want) __builtin_amdgcn_s_sleep(2);
        __syncthreads();
    }
    if (threadIdx.x < 64) {   // lane q loads rank q's partials (the loads overlap), lane 0 adds them in order
        double v[NV];
#pragma unroll
        for (int j = 0; j < NV; j++) v[j] = threadIdx.x < (unsigned)rs.n ? ld_sys(rs.mb + 4 * threadIdx.x + j) : 0.0;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            double s = 0;
            for (int q = 0; q < rs.n; q++) s += __shfl(v[j], q, 64);
            if (threadIdx.x == 0) sh[j] = s;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; j++) out[j] = sh[j];
    __syncthreads();   // sh may be reused by the caller's block sums
}
