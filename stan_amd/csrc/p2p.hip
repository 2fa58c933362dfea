// p2p.hip -- peer-to-peer exchanges of the one-process multi-GPU handle (stan_hip_init_multi):
// the resources (mailboxes, arrival counters), the halo write, the stream waits and the host
// barrier.  The reference is one process on one device (Solver.cs:18-69); the sharded CG is this
// library's own design and SURVEY.md sections 5 / 8e ask for exchanges that cost no collective
// launch: "avoid ring all-reduce for 16-byte payloads", "peer-mapped buffers".
//
// What an iteration of the sharded classic loop enqueues with STAN_OPT_COMM_P2P on (per rank):
//   k_pack_p2p   boundary rows of p -> the neighbours' gather vectors + arrival count
//   k_spmv       interior slices (side stream), then [wait: halo count] boundary slices; the block
//                that finishes the p.Ap sum stores this rank's partial into EVERY rank's mailbox
//   [wait: reduction count]  k_step   (adds the N partials in rank order; r.r and merit likewise)
//   [wait: reduction count]  k_update
// = 4 kernels, 3 stream waits, no RCCL launch (RCCL path: 4 kernels + 3 collective launches).
// The waits are a one-wave polling kernel on the arrival counter (fine-grained device memory of the waiting rank);
// STAN_P2P_WAIT_MODE=0 selects hipStreamWaitValue64 on the same counter instead (stan_p2p_create).
// One rule follows from the waits: NO hipFree inside a peer-to-peer solve -- hipFree waits for every stream of
// the device, and when ranks share a device (the test topology) the other ranks' streams are waiting for this
// rank's future exchanges (stan_ctx::defer_frees; found as stalls in a quarter of the three- and four-rank runs).
#include <unistd.h>

#include <chrono>
#include <cstdio>

#include "internal.h"
#include "p2p_device.h"

namespace {

constexpr unsigned long long RELEASE_ALL = 1ULL << 62;   // stan_p2p_abort: every wait is satisfied

// boundary rows of `vec` into the halo regions of the neighbours' vectors, then one arrival count
// per neighbour by the block that finishes last (every block has waited for its stores first)
struct pack_args {
    int64_t stot;                       // rows to send, grouped by neighbour
    const int32_t *rows;
    const double *vec;
    int32_t n_nbr;
    int64_t send_off[STAN_P2P_MAXR];    // [n_nbr + 1]
    double *dst[STAN_P2P_MAXR];         // neighbour i's halo segment for this rank
    unsigned long long *sig[STAN_P2P_MAXR];
    unsigned long long *tick;
};
__global__ void __launch_bounds__(256) k_pack_p2p(pack_args a) {
    __shared__ int sh_last;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < 3 * a.stot; t += stride) {
        const int64_t i = t / 3;
        int nb = 0;
        while (nb + 1 < a.n_nbr && i >= a.send_off[nb + 1]) nb++;
        const int c = (int)(t - 3 * i);
        st_sys(a.dst[nb] + (t - 3 * a.send_off[nb]), a.vec[3 * (int64_t)a.rows[i] + c]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);
        const unsigned long long t = __hip_atomic_fetch_add(a.tick, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh_last = t == (unsigned long long)gridDim.x - 1;
    }
    __syncthreads();
    if (!sh_last) return;
    if (threadIdx.x == 0) __hip_atomic_store(a.tick, 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)threadIdx.x < a.n_nbr && a.sig[threadIdx.x]) {
        __atomic_thread_fence(__ATOMIC_RELEASE);
        __hip_atomic_fetch_add(a.sig[threadIdx.x], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// wait_mode 1: one wavefront polls the counter (a device flag in fine-grained memory)
__global__ void k_wait_flag(const unsigned long long *flag, unsigned long long want) {
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) __builtin_amdgcn_s_sleep(4);
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
}

// An arrival counter: 128 B of fine-grained DEVICE memory of the rank that waits on it (arrivals are atomics over
// xGMI into the waiter's own HBM, the waiter polls local memory).  Both kinds of wait take it, within one process
// and through an IPC mapping: profiles/r03/waitvalue_probe_*.txt, ipc_probe_two_processes_gpu0.txt.  (The first
// version used HSA signal memory, which hipStreamWaitValue64's documentation asks for; its value lives in HOST
// memory: every arrival a GPU atomic across PCIe.)
int alloc_counter(stan_p2p *, unsigned long long **out) {
    *out = nullptr;
    if (hipExtMallocWithFlags((void **)out, 128, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); return STAN_E_ALLOC; }
    return hipMemset(*out, 0, 128) == hipSuccess ? STAN_OK : STAN_E_HIP;
}

}  // namespace

// Decides whether the devices can exchange peer to peer and how the streams will wait.
int stan_p2p_create(stan_p2p **out, const std::vector<int> &devices, std::string *err) {
    *out = nullptr;
    const int n = (int)devices.size();
    if (n < 2 || n > STAN_P2P_MAXR) { *err = "peer-to-peer exchanges need 2.." + std::to_string(STAN_P2P_MAXR) + " ranks"; return STAN_E_UNSUPPORTED; }
    for (int a = 0; a < n; a++)
        for (int b = 0; b < n; b++) {
            if (devices[a] == devices[b]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) != hipSuccess || !can) {
                (void)hipGetLastError();
                *err = "device " + std::to_string(devices[a]) + " cannot access device " + std::to_string(devices[b]);
                return STAN_E_UNSUPPORTED;
            }
        }
    stan_p2p *pp = new stan_p2p();
    pp->n = n;
    pp->rk.resize((size_t)n);
    for (int r = 0; r < n; r++) pp->rk[(size_t)r].device = devices[r];
    // How a stream waits for an arrival count.  1 (default): a one-wave polling kernel (k_wait_flag): nothing but
    // documented HIP.  0 (STAN_P2P_WAIT_MODE=0): hipStreamWaitValue64 on the counter -- no wavefront spins, as fast
    // (2.3-4.2 us against 2.4 us, profiles/r03/waitvalue_probe_*.txt), and it works on plain device memory on this
    // stack, but its documentation asks for signal memory (host-resident: every arrival an atomic across PCIe).
    // Both pass the same repetitions (24 of 24 each, profiles/r03/p2p_hang_hunt/).
    // 2: like 1 for the halo, but the REDUCTIONS enqueue no wait at all -- the consuming kernel polls the counter
    // itself (red_get: the "single-kernel one-shot" of SURVEY.md section 5; two launches less per iteration).  For
    // one device per rank: a consumer spinning in all its workgroups must not share its device with the producer
    // it waits for (the tests do that with systems small enough to be resident together).
    pp->wait_mode = 1;
    if (const char *m = getenv("STAN_P2P_WAIT_MODE")) {
        pp->wait_mode = atoi(m) == 2 ? 2 : atoi(m) ? 1 : 0;
        for (int r = 0; r < n && pp->wait_mode == 0; r++) {
            int can_wait = 0;
            if (hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, devices[r]) != hipSuccess || !can_wait) {
                (void)hipGetLastError();
                pp->wait_mode = 1;
            }
        }
    }
    *out = pp;
    return STAN_OK;
}

// rank's own thread, its device current: peer access, mailbox, counters
int stan_p2p_rank_setup(stan_p2p *pp, int rank, std::string *err) {
    stan_p2p::rank_res &me = pp->rk[(size_t)rank];
    for (int q = 0; q < pp->n; q++) {
        if (pp->rk[(size_t)q].device == me.device) continue;
        const hipError_t e = hipDeviceEnablePeerAccess(pp->rk[(size_t)q].device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { *err = std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e); return STAN_E_HIP; }
        (void)hipGetLastError();
    }
    const size_t mb = (size_t)STAN_P2P_RING * pp->n * 4 * sizeof(double);
    if (hipExtMallocWithFlags((void **)&me.mbox, mb < 4096 ? 4096 : mb, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        *err = "peer-to-peer mailbox: no fine-grained device memory";
        return STAN_E_ALLOC;
    }
    if (hipMemset(me.mbox, 0, mb < 4096 ? 4096 : mb) != hipSuccess) { *err = "peer-to-peer mailbox: hipMemset failed"; return STAN_E_HIP; }
    for (int s = 0; s < STAN_P2P_RING; s++) {
        int rc = alloc_counter(pp, &me.sig_red[s]);
        if (rc == STAN_OK) rc = alloc_counter(pp, &me.sig_halo[s]);
        if (rc != STAN_OK) { *err = "peer-to-peer arrival counters: allocation failed"; return rc; }
    }
    if (hipMalloc((void **)&me.d_dev, sizeof(stan_p2p_dev)) != hipSuccess || hipMalloc((void **)&me.d_tick, 128) != hipSuccess ||
        hipMemset(me.d_tick, 0, 128) != hipSuccess) {
        (void)hipGetLastError();
        *err = "peer-to-peer tables: hipMalloc failed";
        return STAN_E_ALLOC;
    }
    // does the stream wait take this kind of memory here?  (a wait that is already satisfied)
    if (pp->wait_mode == 0 &&
        (hipStreamWaitValue64(nullptr, me.sig_red[0], 0, hipStreamWaitValueGte, ~0ULL) != hipSuccess ||
         hipStreamSynchronize(nullptr) != hipSuccess)) {
        (void)hipGetLastError();
        pp->wait_mode = 1;   // (every rank polls then: the flag is shared; a racing reader sees 0 or 1, both work)
    }
    return STAN_OK;
}

// after EVERY rank's setup: the table of all ranks' mailboxes and counters goes to this rank's device
int stan_p2p_rank_finish(stan_p2p *pp, int rank, std::string *err) {
    stan_p2p_dev t{};
    t.n = pp->n;
    t.me = rank;
    for (int q = 0; q < pp->n; q++) {
        t.mbox[q] = pp->rk[(size_t)q].mbox;
        for (int s = 0; s < STAN_P2P_RING; s++) {
            t.sig_red[q][s] = pp->rk[(size_t)q].sig_red[s];
            t.sig_halo[q][s] = pp->rk[(size_t)q].sig_halo[s];
        }
    }
    if (hipMemcpy(pp->rk[(size_t)rank].d_dev, &t, sizeof(t), hipMemcpyHostToDevice) != hipSuccess) {
        *err = "peer-to-peer tables: upload failed";
        return STAN_E_HIP;
    }
    return STAN_OK;
}

void stan_p2p_rank_release(stan_p2p *pp, int rank) {
    stan_p2p::rank_res &me = pp->rk[(size_t)rank];
    if (me.mbox) hipFree(me.mbox);
    for (int s = 0; s < STAN_P2P_RING; s++) {
        if (me.sig_red[s]) hipFree(me.sig_red[s]);
        if (me.sig_halo[s]) hipFree(me.sig_halo[s]);
        me.sig_red[s] = me.sig_halo[s] = nullptr;
    }
    if (me.d_dev) hipFree(me.d_dev);
    if (me.d_tick) hipFree(me.d_tick);
    me.mbox = nullptr; me.d_dev = nullptr; me.d_tick = nullptr;
}

void stan_p2p_destroy(stan_p2p *pp) { delete pp; }

// A rank failed: every stream wait of every rank is satisfied from now on (the counters jump past any
// value a solve can ask for), host barriers return STAN_E_COMM, and the CG loops of the surviving
// ranks see `broken` at their next status poll.
void stan_p2p_abort(stan_p2p *pp) {
    if (!pp) return;
    pp->broken.store(true);
    int dev0 = -1;   // this runs on the HOST APPLICATION's thread (multi.hip: run_all): its current device is kept
    if (hipGetDevice(&dev0) != hipSuccess) { (void)hipGetLastError(); dev0 = -1; }
    for (stan_p2p::rank_res &r : pp->rk)
        for (int s = 0; s < STAN_P2P_RING; s++)
            for (unsigned long long *c : {r.sig_red[s], r.sig_halo[s]}) {
                if (!c) continue;
                const unsigned long long v = RELEASE_ALL;
                (void)hipSetDevice(r.device);
                (void)hipMemcpy(c, &v, 8, hipMemcpyHostToDevice);
            }
    if (dev0 >= 0) (void)hipSetDevice(dev0);
    std::lock_guard<std::mutex> lk(pp->m);
    pp->cv.notify_all();
}

double stan_p2p_stall_seconds() {
    if (const char *e = getenv("STAN_P2P_STALL_S")) {
        const double v = atof(e);
        if (v > 0) return v;
    }
    return 120.0;
}

// The process-per-GPU form has no group thread that could abort on its behalf (stan_p2p_abort belongs to
// multi.hip's run_all): the rank releases ITSELF.  The counters a rank waits on are its own local memory, so the
// polling wavefronts of k_wait_flag / red_get (and a hipStreamWaitValue64) see the value at once and the queue drains.
void stan_p2p_release_own(stan_ctx *ctx) {
    stan_p2p *pp = ctx->p2p;
    if (!pp) return;
    if (!pp->ipc) { stan_p2p_abort(pp); return; }   // one process: every rank's waits, as run_all would
    pp->broken.store(true);
    stan_p2p::rank_res &me = pp->rk[(size_t)pp->ipc_me];
    (void)hipSetDevice(ctx->device);
    for (int s = 0; s < STAN_P2P_RING; s++)
        for (unsigned long long *c : {me.sig_red[s], me.sig_halo[s]}) {
            if (!c) continue;
            const unsigned long long v = RELEASE_ALL;
            (void)hipMemcpy(c, &v, 8, hipMemcpyHostToDevice);   // (the context's streams are non-blocking streams)
        }
    (void)hipGetLastError();
}

// Where does a stalled exchange stand?  Every rank's call counts, the arrivals each of its counters must have
// reached and the arrivals it has (read from another thread: the ranks' streams are non-blocking streams).
void stan_p2p_dump(stan_p2p *pp, FILE *f) {
    if (!pp) return;
    int dev0 = -1;   // caller's current device is kept (see stan_p2p_abort)
    if (hipGetDevice(&dev0) != hipSuccess) { (void)hipGetLastError(); dev0 = -1; }
    fprintf(f, "peer-to-peer state (%d ranks, %s, wait mode %d, broken %d)\n", pp->n, pp->ipc ? "IPC" : "one process", pp->wait_mode,
            (int)pp->broken.load());
    for (int r = 0; r < pp->n; r++) {
        stan_p2p::rank_res &k = pp->rk[(size_t)r];
        if (pp->ipc && r != pp->ipc_me) continue;
        (void)hipSetDevice(k.device);
        fprintf(f, "  rank %d: reductions issued %lld, halo exchanges issued %lld\n", r, (long long)k.red_calls, (long long)k.halo_calls);
        for (int s = 0; s < STAN_P2P_RING; s++) {
            unsigned long long a = 0, b = 0;
            if (k.sig_red[s]) (void)hipMemcpy(&a, k.sig_red[s], 8, hipMemcpyDeviceToHost);
            if (k.sig_halo[s]) (void)hipMemcpy(&b, k.sig_halo[s], 8, hipMemcpyDeviceToHost);
            fprintf(f, "    slot %d: reduction arrivals %llu of %llu expected%s | halo arrivals %llu of %llu expected%s\n", s, a,
                    k.red_expect[s], a < k.red_expect[s] ? "  <-- waiting" : "", b, k.halo_expect[s], b < k.halo_expect[s] ? "  <-- waiting" : "");
        }
    }
    if (dev0 >= 0) (void)hipSetDevice(dev0);
    fflush(f);
}

int stan_p2p_barrier(stan_p2p *pp) {
    std::unique_lock<std::mutex> lk(pp->m);
    if (pp->broken.load()) return STAN_E_COMM;
    const long gen = pp->generation;
    if (++pp->arrived == pp->n) {
        pp->arrived = 0;
        pp->generation++;
        pp->cv.notify_all();
        return STAN_OK;
    }
    // bounded: a peer that never arrives (it failed before the solve) must not hang this rank
    const bool ok = pp->cv.wait_for(lk, std::chrono::seconds(120), [&] { return pp->generation != gen || pp->broken.load(); });
    if (!ok || pp->broken.load()) { pp->broken.store(true); return STAN_E_COMM; }
    return STAN_OK;
}

// ---- called from the CG (cg.hip) on a rank context --------------------------------------------------

// stream-ordered wait until counter `c` has reached `want` arrivals (cumulative over its uses)
static int p2p_wait(stan_ctx *ctx, unsigned long long *c, unsigned long long want) {
    stan_p2p *pp = ctx->p2p;
    if (pp->broken.load()) { ctx->err = "peer-to-peer exchange: a peer rank failed"; return STAN_E_COMM; }
    if (pp->wait_mode == 0) HIPCHK(ctx, hipStreamWaitValue64(ctx->stream, c, want, hipStreamWaitValueGte, ~0ULL));
    else hipLaunchKernelGGL(k_wait_flag, dim3(1), dim3(64), 0, ctx->stream, (const unsigned long long *)c, want);
    return STAN_OK;
}

// the slot the NEXT reduction of this rank uses (cg.hip passes it to the producing kernel), and the wait
// for it; every rank makes the same sequence of calls
int stan_p2p_reduce_slot(stan_ctx *ctx) { return (int)(ctx->p2p->rk[(size_t)ctx->rank].red_calls % STAN_P2P_RING); }
// ctr / want non-null and wait mode 2: nothing is enqueued, the consuming kernels poll *ctr >= *want themselves
int stan_p2p_reduce_wait(stan_ctx *ctx, const unsigned long long **ctr, unsigned long long *want) {
    stan_p2p::rank_res &me = ctx->p2p->rk[(size_t)ctx->rank];
    const int slot = (int)(me.red_calls++ % STAN_P2P_RING);
    me.red_expect[slot] += (unsigned long long)ctx->p2p->n;   // every rank counts once per reduction
    if (ctr) *ctr = nullptr;
    if (ctr && want && ctx->p2p->wait_mode == 2) {
        if (ctx->p2p->broken.load()) { ctx->err = "peer-to-peer exchange: a peer rank failed"; return STAN_E_COMM; }
        *ctr = me.sig_red[slot];
        *want = me.red_expect[slot];
        return STAN_OK;
    }
    return p2p_wait(ctx, me.sig_red[slot], me.red_expect[slot]);
}
const stan_p2p_dev *stan_p2p_table(stan_ctx *ctx) { return ctx->p2p->rk[(size_t)ctx->rank].d_dev; }
const double *stan_p2p_mailbox(stan_ctx *ctx, int slot) {
    return ctx->p2p->rk[(size_t)ctx->rank].mbox + (size_t)slot * ctx->p2p->n * 4;
}

// ---- process-per-GPU form: the same exchanges between PROCESSES, peers mapped through HIP IPC -----------
namespace {
constexpr size_t CTR_STRIDE = 128;   // one counter per 128-B line
size_t ipc_mbox_bytes(int n) { return ((size_t)STAN_P2P_RING * n * 4 * sizeof(double) + 255) & ~(size_t)255; }
size_t ipc_block_bytes(int n) { return ipc_mbox_bytes(n) + 2 * STAN_P2P_RING * CTR_STRIDE; }
void ipc_carve(stan_p2p::rank_res &r, void *block, int n) {
    r.mbox = (double *)block;
    unsigned char *c = (unsigned char *)block + ipc_mbox_bytes(n);
    for (int s = 0; s < STAN_P2P_RING; s++) {
        r.sig_red[s] = (unsigned long long *)(c + (size_t)s * CTR_STRIDE);
        r.sig_halo[s] = (unsigned long long *)(c + (size_t)(STAN_P2P_RING + s) * CTR_STRIDE);
    }
}
struct ipc_hello { hipIpcMemHandle_t h; int32_t device, pid; };
struct ipc_vectors {   // what a rank publishes at the start of a solve
    hipIpcMemHandle_t h[5];
    int64_t nloc;
    int32_t n_nbr, pad;
    int32_t nbr[STAN_P2P_MAXR];
    int64_t recv_off[STAN_P2P_MAXR + 1];
};
}  // namespace

int stan_p2p_ipc_setup(stan_ctx *ctx) {
    if (ctx->p2p) return STAN_OK;
    const int n = ctx->nranks;
    if (n < 2 || n > STAN_P2P_MAXR) { ctx->err = "peer-to-peer exchanges need 2.." + std::to_string(STAN_P2P_MAXR) + " ranks"; return STAN_E_UNSUPPORTED; }
    stan_p2p *pp = new stan_p2p();
    pp->n = n; pp->ipc = true; pp->ipc_me = ctx->rank;
    pp->rk.resize((size_t)n);
    pp->ipc_peer_block.assign((size_t)n, nullptr);
    struct guard { stan_p2p *p; stan_ctx *c; bool ok = false; ~guard() { if (!ok) { c->p2p = p; stan_p2p_ipc_release(c); } } } g{pp, ctx};
    stan_p2p::rank_res &me = pp->rk[(size_t)ctx->rank];
    me.device = ctx->device;
    // counters are plain fine-grained device memory here (signal memory cannot be shared between processes);
    // hipStreamWaitValue64 takes them on this stack (profiles/r03/waitvalue_probe_*.txt) -- checked below
    const size_t bytes = ipc_block_bytes(n);
    if (hipExtMallocWithFlags(&pp->ipc_block, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        ctx->err = "peer-to-peer set-up: no fine-grained device memory";
        return STAN_E_ALLOC;
    }
    HIPCHK(ctx, hipMemset(pp->ipc_block, 0, bytes));
    ipc_carve(me, pp->ipc_block, n);
    pp->wait_mode = 1;   // polling kernel (see stan_p2p_create); STAN_P2P_WAIT_MODE=0: hipStreamWaitValue64 where it is accepted
    if (const char *m = getenv("STAN_P2P_WAIT_MODE")) {
        if (atoi(m) == 2) pp->wait_mode = 2;
        int can_wait = 0;
        if (atoi(m) == 0 && hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, ctx->device) == hipSuccess && can_wait &&
            hipStreamWaitValue64(ctx->stream, me.sig_red[0], 0, hipStreamWaitValueGte, ~0ULL) == hipSuccess &&
            hipStreamSynchronize(ctx->stream) == hipSuccess)
            pp->wait_mode = 0;
        (void)hipGetLastError();
    }
    // every rank must end up in the same mode only for its own waits: no agreement needed
    ipc_hello mine{};
    if (hipIpcGetMemHandle(&mine.h, pp->ipc_block) != hipSuccess) {
        ctx->err = std::string("peer-to-peer set-up: hipIpcGetMemHandle: ") + hipGetErrorString(hipGetLastError());
        return STAN_E_HIP;
    }
    mine.device = ctx->device; mine.pid = (int32_t)getpid();
    std::vector<ipc_hello> all((size_t)n);
    STANCHK(stan_comm_allgather_bytes(ctx, &mine, sizeof(mine), all.data()));
    for (int q = 0; q < n; q++) {
        if (q == ctx->rank) continue;
        pp->rk[(size_t)q].device = all[(size_t)q].device;
        if (all[(size_t)q].pid == mine.pid) { ctx->err = "peer-to-peer set-up: two ranks in one process: use stan_hip_init_multi"; return STAN_E_UNSUPPORTED; }
        void *b = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&b, all[(size_t)q].h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            ctx->err = "peer-to-peer set-up: hipIpcOpenMemHandle(rank " + std::to_string(q) + "): " + hipGetErrorString(e);
            return STAN_E_HIP;
        }
        pp->ipc_peer_block[(size_t)q] = b;
        ipc_carve(pp->rk[(size_t)q], b, n);
    }
    if (hipMalloc((void **)&me.d_dev, sizeof(stan_p2p_dev)) != hipSuccess || hipMalloc((void **)&me.d_tick, 128) != hipSuccess ||
        hipMemset(me.d_tick, 0, 128) != hipSuccess) {
        (void)hipGetLastError();
        ctx->err = "peer-to-peer tables: hipMalloc failed";
        return STAN_E_ALLOC;
    }
    std::string why;
    if (stan_p2p_rank_finish(pp, ctx->rank, &why) != STAN_OK) { ctx->err = why; return STAN_E_HIP; }
    g.ok = true;
    ctx->p2p = pp;
    return STAN_OK;
}

void stan_p2p_ipc_release(stan_ctx *ctx) {
    stan_p2p *pp = ctx->p2p;
    if (!pp || !pp->ipc) return;
    hipSetDevice(ctx->device);
    for (auto &kv : pp->ipc_open) if (kv.second) hipIpcCloseMemHandle(kv.second);
    for (void *b : pp->ipc_peer_block) if (b) hipIpcCloseMemHandle(b);
    stan_p2p::rank_res &me = pp->rk[(size_t)pp->ipc_me];
    if (me.d_dev) hipFree(me.d_dev);
    if (me.d_tick) hipFree(me.d_tick);
    if (pp->ipc_block) hipFree(pp->ipc_block);
    (void)hipGetLastError();
    delete pp;
    ctx->p2p = nullptr;
    ctx->comm_p2p = false;
}

// A peer that re-allocated its vectors (another K size, a placement search that moved them, a new d_scale per matrix:
// bench.py assembles a fresh K every step) publishes new handles; the mappings of the old ones would pin the peer's
// freed memory for the life of the context.  Closed at the END of a solve, where the deferred frees are released too
// (hipIpcCloseMemHandle may wait for the device like hipFree: stan_ctx::defer_frees).
void stan_p2p_ipc_trim(stan_ctx *ctx) {
    stan_p2p *pp = ctx->p2p;
    if (!pp || !pp->ipc) return;
    for (auto it = pp->ipc_open.begin(); it != pp->ipc_open.end();) {
        if (pp->ipc_live.count(it->first)) { ++it; continue; }
        if (it->second) (void)hipIpcCloseMemHandle(it->second);
        it = pp->ipc_open.erase(it);
    }
    (void)hipGetLastError();
}

// IPC form of the publication: handles of my five vectors + my halo plan to everybody (a collective: it
// also is the barrier), the peers' vectors mapped (once per allocation: cached by handle)
static int ipc_publish_vectors(stan_ctx *ctx, const stan_matrix *K, double *const vec[5]) {
    stan_p2p *pp = ctx->p2p;
    const int n = pp->n;
    ipc_vectors mine{};
    for (int i = 0; i < 5; i++)
        if (hipIpcGetMemHandle(&mine.h[i], vec[i]) != hipSuccess) {
            ctx->err = std::string("peer-to-peer halo: hipIpcGetMemHandle: ") + hipGetErrorString(hipGetLastError());
            return STAN_E_HIP;
        }
    mine.nloc = K->nloc;
    mine.n_nbr = (int32_t)K->nbr.size();
    for (size_t i = 0; i < K->nbr.size(); i++) mine.nbr[i] = K->nbr[i];
    for (size_t i = 0; i < K->recv_off.size() && i <= (size_t)STAN_P2P_MAXR; i++) mine.recv_off[i] = K->recv_off[i];
    std::vector<ipc_vectors> all((size_t)n);
    STANCHK(stan_comm_allgather_bytes(ctx, &mine, sizeof(mine), all.data()));
    pp->ipc_live.clear();
    for (int q = 0; q < n; q++) {
        stan_p2p::rank_res &r = pp->rk[(size_t)q];
        const ipc_vectors &v = all[(size_t)q];
        r.nloc = v.nloc;
        r.nbr.assign(v.nbr, v.nbr + v.n_nbr);
        r.recv_off.assign(v.recv_off, v.recv_off + v.n_nbr + 1);
        if (q == ctx->rank) { for (int i = 0; i < 5; i++) r.vec[i] = vec[i]; continue; }
        // only the vectors of my neighbours are ever written to
        bool nbr_of_mine = false;
        for (int x : K->nbr) nbr_of_mine |= x == q;
        for (int i = 0; i < 5; i++) {
            r.vec[i] = nullptr;
            if (!nbr_of_mine) continue;
            const std::string key((const char *)&v.h[i], sizeof(hipIpcMemHandle_t));
            pp->ipc_live.insert(key);
            auto it = pp->ipc_open.find(key);
            if (it == pp->ipc_open.end()) {
                void *m = nullptr;
                const hipError_t e = hipIpcOpenMemHandle(&m, v.h[i], hipIpcMemLazyEnablePeerAccess);
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    ctx->err = "peer-to-peer halo: hipIpcOpenMemHandle(rank " + std::to_string(q) + "): " + hipGetErrorString(e);
                    return STAN_E_HIP;
                }
                it = pp->ipc_open.emplace(key, m).first;
            }
            r.vec[i] = (double *)it->second;
        }
    }
    return STAN_OK;
}

// what my neighbours need to know before they write into my vectors; host barrier behind it
int stan_p2p_publish_vectors(stan_ctx *ctx, const stan_matrix *K, double *const vec[5]) {
    stan_p2p *pp = ctx->p2p;
    stan_p2p::rank_res &me = pp->rk[(size_t)ctx->rank];
    if ((int)K->nbr.size() >= STAN_P2P_MAXR) { ctx->err = "peer-to-peer halo: too many neighbour ranks"; return STAN_E_UNSUPPORTED; }
    if (pp->ipc) return ipc_publish_vectors(ctx, K, vec);
    for (int i = 0; i < 5; i++) me.vec[i] = vec[i];
    me.nloc = K->nloc;
    me.nbr = K->nbr;
    me.recv_off = K->recv_off;
    const int rc = stan_p2p_barrier(pp);
    if (rc != STAN_OK) ctx->err = "peer-to-peer exchange: a peer rank failed or never arrived";
    return rc;
}

// owned boundary rows of d_vec (one of the five published vectors) into the neighbours' halo regions,
// then wait until my own halo has been filled by all of mine
int stan_p2p_halo_exchange(stan_ctx *ctx, stan_matrix *K, double *d_vec) {
    stan_p2p *pp = ctx->p2p;
    stan_p2p::rank_res &me = pp->rk[(size_t)ctx->rank];
    if (K->nbr.empty()) return STAN_OK;
    int id = -1;
    for (int i = 0; i < 5; i++) if (me.vec[i] == d_vec) id = i;
    if (id < 0) { ctx->err = "peer-to-peer halo: vector was not published"; return STAN_E_ARG; }
    const int64_t h = me.halo_calls++;
    const int slot = (int)(h % STAN_P2P_RING);
    pack_args a{};
    a.stot = K->send_off.back();
    a.rows = K->d_send_rows;
    a.vec = d_vec;
    a.n_nbr = (int)K->nbr.size();
    a.tick = me.d_tick;
    int n_in = 0;
    for (size_t i = 0; i < K->nbr.size(); i++) {
        const int q = K->nbr[i];
        const stan_p2p::rank_res &pq = pp->rk[(size_t)q];
        a.send_off[i] = K->send_off[i];
        // my segment in q's halo region: q lists me as its j-th neighbour
        int64_t off = -1;
        for (size_t j = 0; j < pq.nbr.size(); j++) if (pq.nbr[j] == ctx->rank) off = pq.recv_off[j];
        const int64_t ns = K->send_off[i + 1] - K->send_off[i];
        if (ns > 0 && off < 0) { ctx->err = "peer-to-peer halo: neighbour lists are not symmetric"; return STAN_E_COMM; }
        a.dst[i] = ns > 0 ? pq.vec[id] + 3 * (pq.nloc + off) : nullptr;
        a.sig[i] = ns > 0 ? pq.sig_halo[slot] : nullptr;
        if (K->recv_off[i + 1] - K->recv_off[i] > 0) n_in++;
    }
    a.send_off[K->nbr.size()] = K->send_off.back();
    if (a.stot > 0) {
        int64_t b = (3 * a.stot + 255) / 256;
        if (b > 1024) b = 1024;
        hipLaunchKernelGGL(k_pack_p2p, dim3((unsigned)b), dim3(256), 0, ctx->stream, a);
    }
    if (n_in > 0) {   // every neighbour that sends to me counts once per exchange
        me.halo_expect[slot] += (unsigned long long)n_in;
        STANCHK(p2p_wait(ctx, me.sig_halo[slot], me.halo_expect[slot]));
    }
    return STAN_OK;
}
