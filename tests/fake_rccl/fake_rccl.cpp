// fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY: a stand-in for the handful of RCCL entry points
// libstan_hip.so resolves with dlopen (comm.hip), implemented over POSIX shared memory and
// host-staged hipMemcpy, so that SEVERAL RANKS CAN SHARE THE ONE GPU of the test box.
// Real RCCL refuses that ("Duplicate GPU detected"), which would leave the sharded CG
// (halo exchange, all-reduces, result gather, two-stream overlap) untested end to end.
// Selected only through the environment variable STAN_RCCL_LIB; the product default is the
// real librccl.so.1.  Every call is synchronous (it drains the stream it is given).
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
struct Comm {
    Header *h;
    char *box;   // nranks mailboxes
    char *pbox;  // nranks x nranks pair mailboxes
    int rank, nranks;
    int local_sense;
    long ncalls;
    char name[64];
    size_t bytes;
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

int flush_group(Comm *c) {
    if (!c) return 5;
    c->ncalls++;
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
    c->bytes = sizeof(Header) + (size_t)nranks * MAILBOX + (size_t)nranks * nranks * PAIRBOX;
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) return 2;
    void *p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 2;
    c->h = (Header *)p;  // a fresh shm segment is zero-filled: arrived = sense = 0
    c->box = (char *)p + sizeof(Header);
    c->pbox = c->box + (size_t)nranks * MAILBOX;
    c->h->nranks = nranks;
    c->h->init.fetch_add(1);
    c->ncalls = 0;
    if (!wait_until(c, [&] { return c->h->init.load() >= nranks; }, "init", -1)) return 2;
    if (!barrier(c, "init barrier")) return 2;
    *comm = c;
    g_comm = c;
    return 0;
}

int ncclCommDestroy(void *comm) {
    Comm *c = (Comm *)comm;
    if (!c->h->aborted.load()) barrier(c, "destroy");
    if (c->rank == 0) shm_unlink(c->name);
    munmap((void *)c->h, c->bytes);
    delete c;
    return 0;
}

int ncclCommAbort(void *comm) {   // like NCCL's: queued and blocked operations of this communicator return
    Comm *c = (Comm *)comm;
    c->h->aborted.store(1);
    return 0;
}

int ncclAllReduce(const void *send, void *recv, size_t count, int dt, int op, void *comm, hipStream_t st) {
    Comm *c = (Comm *)comm;
    if (op != 0 || (dt != 8 && dt != 4)) return 4;
    const size_t bytes = count * 8;
    if (bytes > MAILBOX) return 5;
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
