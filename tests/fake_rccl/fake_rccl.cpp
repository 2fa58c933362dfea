// fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY: a stand-in for the handful of RCCL entry points
// libstan_hip.so resolves with dlopen (comm.hip), implemented over POSIX shared memory and
// host-staged hipMemcpy, so that SEVERAL RANKS CAN SHARE THE ONE GPU of the test box.
// Real RCCL refuses that ("Duplicate GPU detected"), which would leave the sharded CG
// (halo exchange, all-reduces, result gather, two-stream overlap) untested end to end.
// Selected only through the environment variable STAN_RCCL_LIB; the product default is the
// real librccl.so.1.
//
// TWO MODES (round 6):
//   * synchronous (default): every call drains the stream it is given and moves the data on the host.  What runs on
//     ANOTHER stream (the interior product of the two-stream overlap) keeps running: the overlap is real, but its
//     timing is the host's.  FAKE_RCCL_SYNC_DEVICE=1 drains the whole device instead -- a fully serialising
//     transport, behind which a product-side ordering bug hides (tests/test_gpu_sharded.py shows it on a broken build).
//   * FAKE_RCCL_ASYNC=1: ncclSend / ncclRecv / ncclAllReduce are ENQUEUED on the caller's stream like RCCL's --
//     a one-wavefront polling kernel (the wait) + a wide copy kernel per message, the message landing in the
//     RECEIVER'S DEVICE MEMORY (its inbox: the raw pointer between ranks of one process, a HIP IPC mapping between
//     processes; host-registered shared memory only if no IPC mapping can be had), arrival counters in host-registered
//     shared memory written by the stream itself; the host never waits.  What the product forgets to order (a missing
//     hipStreamWaitEvent between the side stream's interior product and the boundary product, say) really
//     runs concurrently.  ncclBroadcast (set-up, the final gather of U) stays host-staged in both modes.
//     FAKE_RCCL_ASYNC_HOST_BOXES=1 forces the host-memory mailboxes (the fallback, compared bit for bit in the tests);
//     FAKE_RCCL_ASYNC_DELAY_US=n makes every message's data land n microseconds after its arrival was announced.
//     Ranks that share one process need a hardware queue per stream (GPU_MAX_HW_QUEUES >= 2 * nranks + 2), as
//     the product's own peer-to-peer path does: a polling kernel must never sit in front of its producer.
//
// Semantics kept from NCCL: collectives (all-reduce, broadcast) are matched by call order and
// need EVERY rank; ncclSend/ncclRecv involve only the two peers (a one-message mailbox per
// ordered pair with produce/consume counters, so ranks without neighbours never take part) and
// pair up per (source, destination) in issue order; ncclAllReduce sums in rank order.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int MAXR = 16;
constexpr size_t MAILBOX = 64u << 20;  // bytes per rank (collectives)
constexpr size_t PAIRBOX = 4u << 20;   // bytes per ordered pair (send/recv); shm pages are lazy
constexpr int MAXOPS = 256;

struct Desc { int peer; int kind; size_t off, bytes; };  // kind 0 = send to peer, 1 = broadcast
struct Header {
    std::atomic<int> init, arrived, sense, aborted;   // aborted: ncclCommAbort by any rank frees every waiter
    int nranks;
    int nops[MAXR];
    Desc ops[MAXR][MAXOPS];
    std::atomic<long> produced[MAXR][MAXR], consumed[MAXR][MAXR];  // [src][dst]
    size_t pair_bytes[MAXR][MAXR];
};
// ---- asynchronous mode: what the stream-ordered kernels read and write (host-registered shared memory) ----
constexpr int ASLOTS = 2;        // messages in flight per ordered pair / all-reduces in flight
constexpr int AR_MAX = 256;      // doubles per all-reduce (the product sends 1..3)
struct alignas(64) Ctr { unsigned long long v; char pad[56]; };   // a counter per cache line
struct AHeader {
    Ctr prod[MAXR][MAXR], cons[MAXR][MAXR];   // [src][dst]: messages written / taken (monotonic)
    Ctr ar_prod[MAXR], ar_cons[MAXR];         // all-reduces a rank has published / finished
    Ctr error;                                // first device-side failure: 1 a wait ran out, 2 message size mismatch
    Ctr abort;                                // ncclCommAbort: every polling kernel returns
    unsigned long long msg_bytes[MAXR][MAXR][ASLOTS];
    unsigned long long ar_box[ASLOTS][MAXR][AR_MAX];   // doubles or int64 as bits
    // every rank's INBOX in device memory (messages land where RCCL's land: in the receiver's HBM, at HBM speed): how the
    // other ranks reach it -- the raw pointer inside one process, a HIP IPC handle between processes
    hipIpcMemHandle_t ipc[MAXR];
    unsigned long long raw[MAXR];
    int pid[MAXR], inbox_ok[MAXR];
};
constexpr size_t PAGE = 4096;
constexpr size_t page_up(size_t b) { return (b + PAGE - 1) / PAGE * PAGE; }

struct Comm {
    Header *h;
    char *box;   // nranks mailboxes
    char *pbox;  // nranks x nranks pair mailboxes
    int rank, nranks;
    int local_sense;
    long ncalls;
    char name[64];
    size_t bytes;
    // asynchronous mode
    bool async = false;
    AHeader *ah = nullptr, *d_ah = nullptr;            // host view / device view of the registered header
    char *abox = nullptr;                              // nranks x nranks x ASLOTS pair mailboxes
    char *d_abox[MAXR][MAXR] = {};                     // device views, registered on first use (the fallback: host mailboxes)
    char *inbox = nullptr;                             // this rank's inbox: [src rank][ASLOTS][PAIRBOX] of device memory
    char *dev_inbox[MAXR] = {};                        // every rank's inbox as this rank reaches it; all null = host mailboxes
    bool inbox_mapped[MAXR] = {};                      // opened through hipIpcOpenMemHandle (to be closed)
    unsigned long long a_sent[MAXR] = {}, a_recvd[MAXR] = {}, ar_calls = 0;
    long delay_us = 0;                                 // FAKE_RCCL_ASYNC_DELAY_US: a message's data lands late
    bool error_reported = false;
    bool sync_device = false;                          // FAKE_RCCL_SYNC_DEVICE: the synchronous mode drains the whole DEVICE
    unsigned *d_tickets = nullptr;                     // device memory: 4096 workgroup tickets (a_ticket)
    unsigned long long n_tickets = 0;
};
struct Op { int kind; const void *src; void *dst; size_t bytes; int peer; hipStream_t st; };  // 0 send 1 recv 2 bcast
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
thread_local Comm *g_comm = nullptr;

// every wait gives up after STALL_S seconds with a dump of what it was waiting for: a protocol
// mismatch between ranks becomes a test failure with a message instead of a hang
constexpr double STALL_S = 60.0;
template <typename P>
bool wait_until(Comm *c, P pred, const char *what, int peer) {
    const auto t0 = std::chrono::steady_clock::now();
    long spins = 0;
    while (!pred()) {
        if (c->h->aborted.load()) return false;
        sched_yield();
        if ((++spins & 0xfff) == 0 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > STALL_S) {
            fprintf(stderr, "fake_rccl: rank %d stalled in %s (peer %d); call #%ld; arrived %d sense %d\n", c->rank,
                    what, peer, c->ncalls, c->h->arrived.load(), c->h->sense.load());
            for (int r = 0; r < c->nranks; r++)
                fprintf(stderr, "  pair %d->%d produced %ld consumed %ld | %d->%d produced %ld consumed %ld\n", c->rank, r,
                        c->h->produced[c->rank][r].load(), c->h->consumed[c->rank][r].load(), r, c->rank,
                        c->h->produced[r][c->rank].load(), c->h->consumed[r][c->rank].load());
            return false;
        }
    }
    return true;
}

bool barrier(Comm *c, const char *what) {
    c->local_sense ^= 1;
    if (c->h->arrived.fetch_add(1) + 1 == c->nranks) {
        c->h->arrived.store(0);
        c->h->sense.store(c->local_sense);
        return true;
    }
    const int want = c->local_sense;
    return wait_until(c, [&] { return c->h->sense.load() == want; }, what, -1);
}
size_t tsize(int dt) { return dt == 1 ? 1 : 8; }  // ncclUint8 = 1; ncclInt64 = 4 and ncclFloat64 = 8 are 8 bytes

char *pair_box(Comm *c, int src, int dst) { return c->pbox + ((size_t)src * c->nranks + dst) * PAIRBOX; }

// ---------------------------------------------------------------------------------------------------------------
// asynchronous mode: kernels.  A polling wave gives up after STALL_S (wall_clock64 ticks at 100 MHz) and leaves
// error = 1, so that a protocol mismatch ends as a failed test and never as a wedged device.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long a_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void a_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ bool a_poll_ge(const Ctr *c, unsigned long long want, AHeader *h) {
    const long long t0 = wall_clock64(), bound = (long long)(STALL_S * 1e8);
    while (a_load(&c->v) < want) {
        if (a_load(&h->abort.v) || a_load(&h->error.v)) return false;   // aborted, or some wait of some rank ran out already
        if (wall_clock64() - t0 > bound) { a_store(&h->error.v, 1); return false; }
        __builtin_amdgcn_s_sleep(16);
    }
    return true;
}
// A message = a WAIT by ONE wavefront (stream-ordered: what follows on the stream starts when it ends) + a copy by many
// workgroups, the last of which (device-scope ticket) publishes at system scope.  Never many workgroups that poll: eight
// ranks sharing one device, each with two neighbours, would fill every CU with pollers and starve the very kernels they
// wait for (round 6: the first fused form hung config 4 on 8 ranks that way).
__global__ void k_a_wait(AHeader *h, const Ctr *c, unsigned long long want, const unsigned long long *size_word,
                         unsigned long long size_want, long delay_us) {
    if (threadIdx.x != 0) return;
    if (!a_poll_ge(c, want, h)) return;
    if (size_word && a_load(size_word) != size_want) { a_store(&h->error.v, 2); return; }
    if (delay_us > 0) {   // FAKE_RCCL_ASYNC_DELAY_US: the data lands that much later -- a consumer that does not wait for
        const long long t0 = wall_clock64();   // the exchange reads the old halo
        while (wall_clock64() - t0 < delay_us * 100) __builtin_amdgcn_s_sleep(8);
    }
}
__global__ void k_a_copy_publish(AHeader *h, char *dst, const char *src, size_t bytes, Ctr *ctr, unsigned long long value,
                                 unsigned long long *size_word, unsigned *ticket) {
    if (a_load(&h->error.v) || a_load(&h->abort.v)) return;   // a failed wait in front of this copy: nothing is published
    const size_t stride = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 15) == 0)
        for (size_t i = t; i < bytes / 16; i += stride) ((uint4 *)dst)[i] = ((const uint4 *)src)[i];
    else
        for (size_t i = t; i < bytes; i += stride) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(ticket, 1u) == gridDim.x - 1) {
        __threadfence_system();
        atomicExch(ticket, 0u);
        if (size_word) a_store(size_word, (unsigned long long)bytes);
        a_store(&ctr->v, value);
    }
}
// all-reduce number k of this communicator, whole: wait until slot k % ASLOTS is free on every rank, publish this
// rank's values, wait for every rank's, add them IN RANK ORDER (what the synchronous mode and the product's
// peer-to-peer path do: the bits agree), say so.  One workgroup; count <= AR_MAX.
__global__ void k_a_allreduce(AHeader *h, int rank, int n, unsigned long long k, const unsigned long long *send,
                              unsigned long long *recv, int count, int is_int) {
    const int t = threadIdx.x, slot = (int)(k % ASLOTS);
    if (k >= ASLOTS && t < n) a_poll_ge(&h->ar_cons[t], k - ASLOTS + 1, h);
    __syncthreads();
    for (int i = t; i < count; i += blockDim.x) a_store(&h->ar_box[slot][rank][i], send[i]);
    __threadfence_system();
    __syncthreads();
    if (t == 0) a_store(&h->ar_prod[rank].v, k + 1);
    if (t < n) a_poll_ge(&h->ar_prod[t], k + 1, h);
    __syncthreads();
    __threadfence_system();
    for (int i = t; i < count; i += blockDim.x) {
        if (is_int) {
            long long s = 0;
            for (int r = 0; r < n; r++) s += (long long)a_load(&h->ar_box[slot][r][i]);
            recv[i] = (unsigned long long)s;
        } else {
            double s = 0;
            for (int r = 0; r < n; r++) s += __longlong_as_double((long long)a_load(&h->ar_box[slot][r][i]));
            recv[i] = (unsigned long long)__double_as_longlong(s);
        }
    }
    __syncthreads();
    if (t == 0) a_store(&h->ar_cons[rank].v, k + 1);
}

char *a_pair_box(Comm *c, int src, int dst) {   // device-visible address of the pair's ASLOTS mailboxes
    if (c->dev_inbox[dst]) return c->dev_inbox[dst] + (size_t)src * ASLOTS * PAIRBOX;   // in the receiver's device memory
    if (!c->d_abox[src][dst]) {                 // fallback: host memory, registered on first use
        char *hp = c->abox + ((size_t)src * c->nranks + dst) * ASLOTS * PAIRBOX;
        void *dp = nullptr;
        if (hipHostRegister(hp, ASLOTS * PAIRBOX, hipHostRegisterMapped) != hipSuccess ||
            hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
            fprintf(stderr, "fake_rccl: hipHostRegister of a pair mailbox failed: %s\n", hipGetErrorString(hipGetLastError()));
            return nullptr;
        }
        c->d_abox[src][dst] = (char *)dp;
    }
    return c->d_abox[src][dst];
}
// a device-side ticket per enqueued message (the last workgroup of a copy publishes): a ring of zeroed counters, each
// put back to zero by the workgroup that drew the last ticket; 4096 messages would have to be in flight for a clash
unsigned *a_ticket(Comm *c) { return c->d_tickets + (c->n_tickets++ & 4095); }
unsigned a_grid(size_t bytes) { size_t b = (bytes / 16 + 255) / 256; return (unsigned)(b < 1 ? 1 : b > 256 ? 256 : b); }

int a_error(Comm *c, const char *where) {   // a device-side failure recorded so far?
    const unsigned long long e = c->ah ? ((volatile Ctr *)&c->ah->error)->v : 0;
    if (!e) return 0;
    if (c->error_reported) return 5;
    c->error_reported = true;
    fprintf(stderr, "fake_rccl (async): rank %d, %s: %s\n", c->rank, where,
            e == 1 ? "a stream-ordered wait ran out (protocol mismatch or a rank that left)" : "message size mismatch");
    return 5;
}

// send / receive of one group, enqueued: nothing here waits on the host
int a_enqueue_p2p(Comm *c) {
    AHeader *d = c->d_ah;
    int rc = a_error(c, "at an exchange");   // a device-side failure recorded so far ends the solve at its next call
    if (rc) return rc;
    for (const Op &o : g_ops) {
        if (o.kind != 0) continue;
        if (o.bytes > PAIRBOX) { fprintf(stderr, "fake_rccl: message larger than the pair mailbox\n"); return 5; }
        char *box = a_pair_box(c, c->rank, o.peer);
        if (!box) return 2;
        const unsigned long long k = c->a_sent[o.peer]++;
        const int slot = (int)(k % ASLOTS);
        if (k >= ASLOTS)   // the slot is free once the receiver has taken message k - ASLOTS
            hipLaunchKernelGGL(k_a_wait, dim3(1), dim3(64), 0, o.st, d, (const Ctr *)&d->cons[c->rank][o.peer], k - ASLOTS + 1,
                               (const unsigned long long *)nullptr, 0ULL, 0L);
        hipLaunchKernelGGL(k_a_copy_publish, dim3(a_grid(o.bytes)), dim3(256), 0, o.st, d, box + (size_t)slot * PAIRBOX,
                           (const char *)o.src, o.bytes, &d->prod[c->rank][o.peer], k + 1,
                           &d->msg_bytes[c->rank][o.peer][slot], a_ticket(c));
    }
    for (const Op &o : g_ops) {
        if (o.kind != 1) continue;
        char *box = a_pair_box(c, o.peer, c->rank);
        if (!box) return 2;
        const unsigned long long k = c->a_recvd[o.peer]++;
        const int slot = (int)(k % ASLOTS);
        hipLaunchKernelGGL(k_a_wait, dim3(1), dim3(64), 0, o.st, d, (const Ctr *)&d->prod[o.peer][c->rank], k + 1,
                           (const unsigned long long *)&d->msg_bytes[o.peer][c->rank][slot], (unsigned long long)o.bytes,
                           c->delay_us);
        hipLaunchKernelGGL(k_a_copy_publish, dim3(a_grid(o.bytes)), dim3(256), 0, o.st, d, (char *)o.dst,
                           (const char *)box + (size_t)slot * PAIRBOX, o.bytes, &d->cons[o.peer][c->rank], k + 1,
                           (unsigned long long *)nullptr, a_ticket(c));
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int flush_group(Comm *c) {
    if (!c) return 5;
    c->ncalls++;
    if (c->async) {
        int rc = a_enqueue_p2p(c);
        if (rc) return rc;
        std::vector<Op> bc;
        for (const Op &o : g_ops) if (o.kind == 2) bc.push_back(o);
        g_ops.swap(bc);
        if (g_ops.empty()) return 0;
        // broadcasts stay host-staged (set-up and the final gather): the code below, on the broadcasts alone;
        // the drain in front of it is also where a device-side failure of the enqueued exchanges surfaces
        for (const Op &o : g_ops) hipStreamSynchronize(o.st);
        if ((rc = a_error(c, "before a broadcast group"))) return rc;
    }
    if (c->sync_device) hipDeviceSynchronize();
    for (const Op &o : g_ops) hipStreamSynchronize(o.st);
    // point-to-point: all sends of the group first (one message in flight per ordered pair),
    // then the receives -- only the two peers of a message ever wait for each other
    for (const Op &o : g_ops) {
        if (o.kind != 0) continue;
        if (o.bytes > PAIRBOX) { fprintf(stderr, "fake_rccl: message larger than the pair mailbox\n"); return 5; }
        std::atomic<long> &pr = c->h->produced[c->rank][o.peer], &co = c->h->consumed[c->rank][o.peer];
        if (!wait_until(c, [&] { return co.load() == pr.load(); }, "send (previous message not taken)", o.peer)) return 5;
        hipMemcpy(pair_box(c, c->rank, o.peer), o.src, o.bytes, hipMemcpyDeviceToHost);
        c->h->pair_bytes[c->rank][o.peer] = o.bytes;
        pr.fetch_add(1);
    }
    for (const Op &o : g_ops) {
        if (o.kind != 1) continue;
        std::atomic<long> &pr = c->h->produced[o.peer][c->rank], &co = c->h->consumed[o.peer][c->rank];
        if (!wait_until(c, [&] { return pr.load() != co.load(); }, "recv", o.peer)) return 5;
        if (c->h->pair_bytes[o.peer][c->rank] != o.bytes) {
            fprintf(stderr, "fake_rccl: rank %d expected %zu bytes from %d, message has %zu\n", c->rank, o.bytes,
                    o.peer, c->h->pair_bytes[o.peer][c->rank]);
            return 5;
        }
        hipMemcpy(o.dst, pair_box(c, o.peer, c->rank), o.bytes, hipMemcpyHostToDevice);
        co.fetch_add(1);
    }
    bool any_bc = false;
    for (const Op &o : g_ops) any_bc |= o.kind == 2;
    if (!any_bc) { g_ops.clear(); return 0; }
    // broadcasts (collective: every rank is here): phase 1, the roots publish
    size_t off = 0;
    int n = 0;
    char *mine = c->box + (size_t)c->rank * MAILBOX;
    for (const Op &o : g_ops) {
        if (o.kind != 2 || o.peer != c->rank) continue;  // not the root of this broadcast
        if (off + o.bytes > MAILBOX || n >= MAXOPS) { fprintf(stderr, "fake_rccl: mailbox overflow\n"); return 5; }
        hipMemcpy(mine + off, o.src, o.bytes, hipMemcpyDeviceToHost);
        c->h->ops[c->rank][n++] = Desc{-1, 1, off, o.bytes};
        off += o.bytes;
    }
    c->h->nops[c->rank] = n;
    if (!barrier(c, "broadcast publish")) return 5;
    // phase 2: pick up, matching in issue order per root
    int taken_bc[MAXR] = {0};
    for (const Op &o : g_ops) {
        if (o.kind != 2) continue;
        if (o.peer == c->rank) {  // root: in place or copy to recv buffer
            if (o.dst != o.src) hipMemcpy(o.dst, o.src, o.bytes, hipMemcpyDeviceToDevice);
            continue;
        }
        const int src = o.peer;
        if (taken_bc[src] >= c->h->nops[src] || c->h->ops[src][taken_bc[src]].bytes != o.bytes) {
            fprintf(stderr, "fake_rccl: rank %d found no matching broadcast from %d\n", c->rank, src);
            return 5;
        }
        const Desc &d = c->h->ops[src][taken_bc[src]++];
        hipMemcpy(o.dst, c->box + (size_t)src * MAILBOX + d.off, o.bytes, hipMemcpyHostToDevice);
    }
    if (!barrier(c, "broadcast done")) return 5;
    g_ops.clear();
    // TEST HOOK FAKE_RCCL_EXIT_AFTER_BCAST_GROUPS="rank:count": that rank's process ends (quietly, code 0) right
    // after its count-th completed group of broadcasts -- i.e. a peer that has published what the others asked
    // for and is then gone (tests/test_gpu_sharded.py: the survivors' peer-to-peer waits must release themselves)
    static int n_bc_groups = 0;
    n_bc_groups++;
    if (const char *e = getenv("FAKE_RCCL_EXIT_AFTER_BCAST_GROUPS")) {
        int r = -1, k = -1;
        if (sscanf(e, "%d:%d", &r, &k) == 2 && r == c->rank && n_bc_groups == k) {
            fprintf(stderr, "fake_rccl: rank %d leaves after broadcast group %d (test hook)\n", r, k);
            fflush(stderr);
            _exit(0);
        }
    }
    return 0;
}

}  // namespace

extern "C" {

int ncclGetUniqueId(void *id) {
    memset(id, 0, 128);
    snprintf((char *)id, 64, "/stan_fake_rccl_%d_%ld", (int)getpid(), (long)random());
    return 0;
}

struct nccl_uid { char internal[128]; };
int ncclCommInitRank(void **comm, int nranks, nccl_uid id, int rank) {
    if (nranks > MAXR) return 4;
    Comm *c = new Comm();
    c->rank = rank; c->nranks = nranks; c->local_sense = 0;
    strncpy(c->name, id.internal, 63);
    const char *am = getenv("FAKE_RCCL_ASYNC");
    c->async = am && atoi(am) != 0;
    const char *sd = getenv("FAKE_RCCL_SYNC_DEVICE");
    c->sync_device = sd && atoi(sd) != 0;
    const size_t sync_bytes = page_up(sizeof(Header) + (size_t)nranks * MAILBOX + (size_t)nranks * nranks * PAIRBOX);
    c->bytes = sync_bytes + (c->async ? page_up(sizeof(AHeader)) + (size_t)nranks * nranks * ASLOTS * PAIRBOX : 0);
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) return 2;
    void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 2;
    c->h = (Header *)p;  // a fresh shm segment is zero-filled: arrived = sense = 0
    c->box = (char *)p + sizeof(Header);
    c->pbox = c->box + (size_t)nranks * MAILBOX;
    c->h->nranks = nranks;
    if (c->async) {   // the counters and the all-reduce slots, visible to this rank's kernels
        c->ah = (AHeader *)((char *)p + sync_bytes);
        c->abox = (char *)c->ah + page_up(sizeof(AHeader));
        void *dp = nullptr;
        if (hipHostRegister(c->ah, page_up(sizeof(AHeader)), hipHostRegisterMapped) != hipSuccess ||
            hipHostGetDevicePointer(&dp, c->ah, 0) != hipSuccess) {
            fprintf(stderr, "fake_rccl: hipHostRegister of the shared header failed: %s\n", hipGetErrorString(hipGetLastError()));
            return 2;
        }
        c->d_ah = (AHeader *)dp;
        if (hipMalloc((void **)&c->d_tickets, 4096 * sizeof(unsigned)) != hipSuccess ||
            hipMemset(c->d_tickets, 0, 4096 * sizeof(unsigned)) != hipSuccess) return 2;
        if (const char *e = getenv("FAKE_RCCL_ASYNC_DELAY_US")) c->delay_us = atol(e);
        // this rank's inbox, and how the others reach it
        c->ah->pid[rank] = (int)getpid();
        c->ah->raw[rank] = 0;
        if (!getenv("FAKE_RCCL_ASYNC_HOST_BOXES") && hipMalloc((void **)&c->inbox, (size_t)nranks * ASLOTS * PAIRBOX) == hipSuccess) {
            if (hipIpcGetMemHandle(&c->ah->ipc[rank], c->inbox) == hipSuccess) c->ah->raw[rank] = (unsigned long long)(uintptr_t)c->inbox;
            else { (void)hipGetLastError(); hipFree(c->inbox); c->inbox = nullptr; }
        } else (void)hipGetLastError();
    }
    c->h->init.fetch_add(1);
    c->ncalls = 0;
    if (!wait_until(c, [&] { return c->h->init.load() >= nranks; }, "init", -1)) return 2;
    if (!barrier(c, "init barrier")) return 2;
    if (c->async) {   // map every rank's inbox; device mailboxes only if EVERY rank reaches every inbox (both ends of a pair must agree)
        bool ok = c->inbox != nullptr;
        char *map[MAXR] = {};
        for (int r = 0; r < nranks && ok; r++) {
            if (!c->ah->raw[r]) { ok = false; break; }
            if (r == rank) map[r] = c->inbox;
            else if (c->ah->pid[r] == (int)getpid()) map[r] = (char *)(uintptr_t)c->ah->raw[r];   // a thread of this process: the pointer itself
            else {
                void *q = nullptr;
                if (hipIpcOpenMemHandle(&q, c->ah->ipc[r], hipIpcMemLazyEnablePeerAccess) == hipSuccess) { map[r] = (char *)q; c->inbox_mapped[r] = true; }
                else { (void)hipGetLastError(); ok = false; }
            }
        }
        c->ah->inbox_ok[rank] = ok ? 1 : 0;
        if (!barrier(c, "inbox exchange")) return 2;
        for (int r = 0; r < nranks; r++) ok = ok && c->ah->inbox_ok[r] == 1;
        for (int r = 0; r < nranks; r++) {
            if (ok) c->dev_inbox[r] = map[r];
            else if (c->inbox_mapped[r]) { hipIpcCloseMemHandle(map[r]); c->inbox_mapped[r] = false; }
        }
        if (rank == 0)
            fprintf(stderr, "fake_rccl: asynchronous mode (stream-ordered send / recv / all-reduce), %d ranks, mailboxes in %s\n", nranks,
                    ok ? "the receivers' device memory" : "host memory (no IPC mapping of the inboxes)");
    }
    *comm = c;
    g_comm = c;
    return 0;
}

int ncclCommDestroy(void *comm) {
    Comm *c = (Comm *)comm;
    if (c->async) {
        hipDeviceSynchronize();
        a_error(c, "at ncclCommDestroy");
    }
    if (!c->h->aborted.load()) barrier(c, "destroy");
    if (c->async) {
        for (int a = 0; a < c->nranks; a++)
            for (int b = 0; b < c->nranks; b++)
                if (c->d_abox[a][b]) hipHostUnregister(c->abox + ((size_t)a * c->nranks + b) * ASLOTS * PAIRBOX);
        hipHostUnregister(c->ah);
        hipFree(c->d_tickets);
        for (int r = 0; r < c->nranks; r++)
            if (c->inbox_mapped[r]) hipIpcCloseMemHandle(c->dev_inbox[r]);
    }
    if (c->rank == 0) shm_unlink(c->name);
    munmap((void *)c->h, c->bytes);
    if (c->inbox) hipFree(c->inbox);   // (after the barrier above: no peer still copies into it)
    delete c;
    return 0;
}

int ncclCommAbort(void *comm) {   // like NCCL's: queued and blocked operations of this communicator return
    Comm *c = (Comm *)comm;
    c->h->aborted.store(1);
    if (c->ah) ((volatile Ctr *)&c->ah->abort)->v = 1;   // the polling kernels of every rank return
    return 0;
}

int ncclAllReduce(const void *send, void *recv, size_t count, int dt, int op, void *comm, hipStream_t st) {
    Comm *c = (Comm *)comm;
    if (op != 0 || (dt != 8 && dt != 4)) return 4;
    const size_t bytes = count * 8;
    if (bytes > MAILBOX) return 5;
    if (c->async) {
        if (count > (size_t)AR_MAX) return 4;
        if (a_error(c, "at an all-reduce")) return 5;
        c->ncalls++;
        hipLaunchKernelGGL(k_a_allreduce, dim3(1), dim3(256), 0, st, c->d_ah, c->rank, c->nranks, c->ar_calls++,
                           (const unsigned long long *)send, (unsigned long long *)recv, (int)count, dt == 4 ? 1 : 0);
        return hipGetLastError() == hipSuccess ? 0 : 2;
    }
    if (c->sync_device) hipDeviceSynchronize();
    hipStreamSynchronize(st);
    c->ncalls++;
    hipMemcpy(c->box + (size_t)c->rank * MAILBOX, send, bytes, hipMemcpyDeviceToHost);
    if (!barrier(c, "allreduce gather")) return 5;
    std::vector<char> out(bytes);
    for (size_t i = 0; i < count; i++) {
        if (dt == 8) {
            double s = 0;
            for (int r = 0; r < c->nranks; r++) s += ((const double *)(c->box + (size_t)r * MAILBOX))[i];
            ((double *)out.data())[i] = s;
        } else {
            int64_t s = 0;
            for (int r = 0; r < c->nranks; r++) s += ((const int64_t *)(c->box + (size_t)r * MAILBOX))[i];
            ((int64_t *)out.data())[i] = s;
        }
    }
    if (!barrier(c, "allreduce done")) return 5;
    hipMemcpy(recv, out.data(), bytes, hipMemcpyHostToDevice);
    return 0;
}

int ncclGroupStart() { g_depth++; return 0; }
int ncclGroupEnd() {
    if (--g_depth > 0) return 0;
    return g_ops.empty() ? 0 : flush_group(g_comm);
}
int ncclSend(const void *buf, size_t count, int dt, int peer, void *comm, hipStream_t st) {
    g_comm = (Comm *)comm;
    g_ops.push_back(Op{0, buf, nullptr, count * tsize(dt), peer, st});
    return g_depth ? 0 : flush_group(g_comm);
}
int ncclRecv(void *buf, size_t count, int dt, int peer, void *comm, hipStream_t st) {
    g_comm = (Comm *)comm;
    g_ops.push_back(Op{1, nullptr, buf, count * tsize(dt), peer, st});
    return g_depth ? 0 : flush_group(g_comm);
}
int ncclBroadcast(const void *send, void *recv, size_t count, int dt, int root, void *comm, hipStream_t st) {
    g_comm = (Comm *)comm;
    g_ops.push_back(Op{2, send, recv, count * tsize(dt), root, st});
    return g_depth ? 0 : flush_group(g_comm);
}
int ncclGetVersion(int *v) { *v = 0; return 0; }   // 0 = "not RCCL": bench.py prints it
int ncclCommCount(void *comm, int *n) { *n = ((Comm *)comm)->nranks; return 0; }
int ncclCommUserRank(void *comm, int *r) { *r = ((Comm *)comm)->rank; return 0; }
const char *ncclGetErrorString(int r) {
    return r == 0 ? "success" : r == 4 ? "fake_rccl: invalid argument" : r == 5 ? "fake_rccl: protocol error" : "fake_rccl: system error";
}

}  // extern "C"
