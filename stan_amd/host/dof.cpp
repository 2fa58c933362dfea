// dof.cpp -- host-side integer steps around the GPU hot path (include/stan_host.h).
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "../../include/stan_host.h"

namespace stan {
int HostThreads();
namespace {
template <typename F>
void par_ranges(int64_t n, int threads, F fn) {
    if (threads <= 1 || n < 65536) { fn((int64_t)0, n); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < threads; t++) th.emplace_back([=] { fn(n * t / threads, n * (t + 1) / threads); });
    fn((int64_t)0, n / threads);
    for (std::thread &x : th) x.join();
}
}  // namespace

// Database.AssignDOF (Database.cs:140-234).
//
// The reference materialises a neighbour list per node (element iteration order x NList
// order, Distinct(), self removed: Database.cs:161-176) and walks a FIFO list that may
// hold a node several times, numbering a node the first time it is popped
// (Database.cs:209-233).  The first occurrence of every node in that list is created when
// the first already-numbered neighbour is processed, so "mark on first push" yields the
// same numbering with a queue of exactly n_nodes entries and no stored neighbour lists:
// neighbours are enumerated on the fly from the node->element incidence (EList order =
// element order with duplicates removed, Node.cs:202-205).
// Round 5: the incidence table (EList as CSR) is built on the host threads -- counted and filled with atomic cursors in
// whatever order the threads arrive, then every node's few entries sorted: ascending element index IS the ElemLib order the
// serial fill produced -- and handed to the caller (eptr_out / elist_out), who needs the same lists for Node.EList.  The
// walk itself stays serial: its order is the result.
int AssignDofCore(int64_t n_nodes, int64_t n_elem, const int32_t *conn, int32_t *node_index_out, int32_t *node_dof_out,
                  std::vector<int64_t> *eptr_out, std::vector<int32_t> *elist_out) {
    if (n_nodes <= 0 || n_elem < 0 || !conn || !node_index_out) return STAN_HOST_E_ARG;
    const int threads = HostThreads();
    const bool trace = getenv("STAN_HOST_TRACE") != nullptr;   // stage times on stderr (diagnosis)
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (trace) fprintf(stderr, "[stan host] AssignDOF %-28s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        t0 = std::chrono::steady_clock::now();
    };
    std::atomic<int> bad{0};
    par_ranges(n_elem * 8, threads, [&](int64_t a, int64_t b) {
        for (int64_t t = a; t < b; t++)
            if (conn[t] < 0 || conn[t] >= n_nodes) { bad.store(1); return; }
    });
    if (bad.load()) return STAN_HOST_E_ARG;
    // EList (CSR), distinct per node: Element.AddElem2Nodes (Element.cs:474-480) in ElemLib order
    std::vector<std::atomic<int32_t>> cnt((size_t)n_nodes);
    par_ranges(n_nodes, threads, [&](int64_t a, int64_t b) { for (int64_t i = a; i < b; i++) cnt[(size_t)i].store(0, std::memory_order_relaxed); });
    auto first_listing = [&](int64_t e, int a) {
        const int32_t nd = conn[e * 8 + a];
        for (int b = 0; b < a; b++) if (conn[e * 8 + b] == nd) return false;
        return true;
    };
    par_ranges(n_elem, threads, [&](int64_t e0, int64_t e1) {
        for (int64_t e = e0; e < e1; e++)
            for (int a = 0; a < 8; a++)
                if (first_listing(e, a)) cnt[(size_t)conn[e * 8 + a]].fetch_add(1, std::memory_order_relaxed);
    });
    std::vector<int64_t> eptr((size_t)n_nodes + 1, 0);
    for (int64_t i = 0; i < n_nodes; i++) eptr[(size_t)i + 1] = eptr[(size_t)i] + cnt[(size_t)i].load(std::memory_order_relaxed);
    std::vector<int32_t> elist((size_t)eptr[(size_t)n_nodes]);
    par_ranges(n_nodes, threads, [&](int64_t a, int64_t b) { for (int64_t i = a; i < b; i++) cnt[(size_t)i].store(0, std::memory_order_relaxed); });
    par_ranges(n_elem, threads, [&](int64_t e0, int64_t e1) {
        for (int64_t e = e0; e < e1; e++)
            for (int a = 0; a < 8; a++)
                if (first_listing(e, a)) {
                    const int32_t nd = conn[e * 8 + a];
                    elist[(size_t)(eptr[(size_t)nd] + cnt[(size_t)nd].fetch_add(1, std::memory_order_relaxed))] = (int32_t)e;
                }
    });
    par_ranges(n_nodes, threads, [&](int64_t a, int64_t b) {
        for (int64_t i = a; i < b; i++) std::sort(elist.begin() + eptr[(size_t)i], elist.begin() + eptr[(size_t)i + 1]);
    });
    lap("incidence table (threads)");
    // Database.cs:178-196: first node (NodeLib order) contained in exactly 1, else 2 ... 6 elements
    int64_t first = -1;
    for (int c = 1; c < 7 && first < 0; c++)
        for (int64_t i = 0; i < n_nodes; i++)
            if (eptr[(size_t)i + 1] - eptr[(size_t)i] == c) { first = i; break; }
    if (first < 0) return STAN_HOST_E_NO_START;

    // The walk.  The queue is the result: node k of it gets index k.  Serial form: pop u, scan its elements (EList order) and
    // their nodes (NList order), append every node seen for the first time.  The queue is a sequence of LEVELS (level k + 1 =
    // what the scan of level k discovers), and within a level the order is "first occurrence in the concatenated scans of
    // the level's nodes" -- which threads can produce without walking one node after the other (round 5): every thread
    // scans a contiguous piece of the level and claims each unseen node with the key (position of the scanning node in the
    // level, offset inside its scan) by an atomic minimum; the smallest key of a node is its first occurrence; a second
    // scan keeps the candidates whose key won, and the pieces' winners, concatenated in piece order, ARE the serial order.
    // Levels narrower than `par_min` nodes are walked serially (the first and last levels of a cube, every level of a small
    // mesh); STAN_HOST_BFS_PAR_MIN overrides (tests: 1).
    std::vector<int32_t> queue((size_t)n_nodes);
    constexpr uint64_t UNSEEN = ~(uint64_t)0;
    std::vector<std::atomic<uint64_t>> state((size_t)n_nodes);   // UNSEEN, a claim key of the level being scanned, or 0 = in the queue
    par_ranges(n_nodes, threads, [&](int64_t a, int64_t b) { for (int64_t i = a; i < b; i++) state[(size_t)i].store(UNSEEN, std::memory_order_relaxed); });
    int64_t par_min = 4096;
    if (const char *e = getenv("STAN_HOST_BFS_PAR_MIN")) par_min = atoll(e) > 0 ? atoll(e) : par_min;
    int64_t head = 0, tail = 0;
    queue[(size_t)tail++] = (int32_t)first;   // Database.cs:209-211; NextNode = Neighbors[FirstNode]
    state[(size_t)first].store(0, std::memory_order_relaxed);
    // persistent workers for the wide levels (a thread per level would cost more than the level).  Between the phases of
    // one wide level a worker spins (yield) for the next phase; through a run of serial, narrow levels -- the first and last
    // levels of a cube, every level of a slender mesh -- it PARKS on a condition variable after a bounded spin instead of
    // burning a core per worker for the whole walk.
    struct Pool {
        int n;
        std::vector<std::thread> th;
        std::atomic<int> phase{0}, arrived{0}, sleepers{0};
        std::atomic<bool> quit{false};
        std::mutex mu;
        std::condition_variable cv;
        std::function<void(int)> job;
        void wake() {   // (phase / quit were stored before this load: a worker either sees them or is counted here)
            if (sleepers.load() > 0) { std::lock_guard<std::mutex> lk(mu); cv.notify_all(); }
        }
        void run(const std::function<void(int)> &f) {   // f(t) on every worker, the caller is worker 0
            job = f;
            arrived.store(0, std::memory_order_relaxed);
            phase.fetch_add(1);
            wake();
            f(0);
            while (arrived.load(std::memory_order_acquire) < n - 1) std::this_thread::yield();
        }
        void stop() {
            quit.store(true);
            wake();
            for (std::thread &x : th) x.join();
        }
    } pool;
    pool.n = threads;
    auto start_pool = [&] {
        for (int t = 1; t < pool.n; t++)
            pool.th.emplace_back([&pool, t] {
                int seen = 0;
                for (;;) {
                    int spins = 0;
                    while (pool.phase.load() == seen) {
                        if (pool.quit.load()) return;
                        if (++spins < 2000) { std::this_thread::yield(); continue; }
                        std::unique_lock<std::mutex> lk(pool.mu);
                        pool.sleepers.fetch_add(1);
                        pool.cv.wait(lk, [&] { return pool.phase.load() != seen || pool.quit.load(); });
                        pool.sleepers.fetch_sub(1);
                    }
                    seen++;
                    pool.job(t);
                    pool.arrived.fetch_add(1, std::memory_order_release);
                }
            });
    };
    std::vector<int64_t> cnt_t((size_t)threads + 1, 0);
    while (tail < n_nodes) {
        if (head >= tail) { pool.stop(); return STAN_HOST_E_DISCONNECTED; }
        const int64_t l0 = head, l1 = tail;   // the level to scan
        if (threads <= 1 || l1 - l0 < par_min) {
            for (int64_t h = l0; h < l1; h++) {
                const int32_t nid = queue[(size_t)h];
                for (int64_t q = eptr[(size_t)nid]; q < eptr[(size_t)nid + 1]; q++) {
                    const int32_t *nl = conn + (int64_t)elist[(size_t)q] * 8;
                    for (int a = 0; a < 8; a++) {
                        const int32_t n = nl[a];
                        if (state[(size_t)n].load(std::memory_order_relaxed) != 0) {
                            state[(size_t)n].store(0, std::memory_order_relaxed);
                            queue[(size_t)tail++] = n;
                        }
                    }
                }
            }
            head = l1;
            continue;
        }
        if (pool.th.empty()) start_pool();
        const int64_t m = l1 - l0;
        auto piece = [&](int t, int64_t *a, int64_t *b) { *a = l0 + m * t / threads; *b = l0 + m * (t + 1) / threads; };
        auto scan = [&](int t, int pass, int64_t out) {   // pass 0: claim; 1: count the winners; 2: write them from `out` on
            int64_t a, b, won = 0;
            piece(t, &a, &b);
            for (int64_t h = a; h < b; h++) {
                const int32_t nid = queue[(size_t)h];
                uint64_t key = ((uint64_t)(h - l0) << 32) + 1;   // (position in the level, offset in the scan) + 1: never 0
                for (int64_t q = eptr[(size_t)nid]; q < eptr[(size_t)nid + 1]; q++) {
                    const int32_t *nl = conn + (int64_t)elist[(size_t)q] * 8;
                    for (int a2 = 0; a2 < 8; a2++, key++) {
                        std::atomic<uint64_t> &st = state[(size_t)nl[a2]];
                        uint64_t cur = st.load(std::memory_order_relaxed);
                        if (pass == 0) {
                            while (cur > key && !st.compare_exchange_weak(cur, key, std::memory_order_relaxed)) {}
                        } else if (cur == key) {
                            if (pass == 2) queue[(size_t)(out + won)] = nl[a2];
                            won++;
                        }
                    }
                }
            }
            return won;
        };
        pool.run([&](int t) { scan(t, 0, 0); });
        pool.run([&](int t) { cnt_t[(size_t)t + 1] = scan(t, 1, 0); });
        cnt_t[0] = 0;
        for (int t = 0; t < threads; t++) cnt_t[(size_t)t + 1] += cnt_t[(size_t)t];
        const int64_t base = tail;
        pool.run([&](int t) { scan(t, 2, base + cnt_t[(size_t)t]); });
        tail = base + cnt_t[(size_t)threads];
        // the winners are in the queue: 0 from now on (a loser's key never equals a winner's, so late readers cannot be confused)
        pool.run([&](int t) {
            for (int64_t i = base + (tail - base) * t / threads; i < base + (tail - base) * (t + 1) / threads; i++)
                state[(size_t)queue[(size_t)i]].store(0, std::memory_order_relaxed);
        });
        head = l1;
    }
    pool.stop();
    par_ranges(n_nodes, threads, [&](int64_t a, int64_t b) { for (int64_t i = a; i < b; i++) node_index_out[queue[(size_t)i]] = (int32_t)i; });
    lap("breadth-first walk");
    if (node_dof_out)
        par_ranges(n_nodes, threads, [&](int64_t a, int64_t b) {
            for (int64_t i = a; i < b; i++) {  // Node.SetDOF, Node.cs:218-223
                node_dof_out[3 * i + 0] = 3 * node_index_out[i];
                node_dof_out[3 * i + 1] = 3 * node_index_out[i] + 1;
                node_dof_out[3 * i + 2] = 3 * node_index_out[i] + 2;
            }
        });
    if (eptr_out) *eptr_out = std::move(eptr);
    if (elist_out) *elist_out = std::move(elist);
    return STAN_HOST_OK;
}
}  // namespace stan

extern "C" {

// Database.AssignDOF (Database.cs:140-234): stan::AssignDofCore above.
int stan_host_assign_dof(int64_t n_nodes, int64_t n_elem, const int32_t *conn,
                         int32_t *node_index_out, int32_t *node_dof_out) {
    return stan::AssignDofCore(n_nodes, n_elem, conn, node_index_out, node_dof_out, nullptr, nullptr);
}

// Solver.cs:104-132
int stan_host_dof_reduction(int64_t n_dof, const int32_t *node_dof, int64_t n_spc,
                            const int32_t *spc_nodes, const double *spc_vals, int32_t *red_out,
                            int64_t *n_fixed_out) {
    if (n_dof < 0 || !red_out || (n_spc > 0 && (!node_dof || !spc_nodes || !spc_vals)))
        return STAN_HOST_E_ARG;
    std::memset(red_out, 0, sizeof(int32_t) * (size_t)n_dof);
    for (int64_t s = 0; s < n_spc; s++)
        for (int d = 0; d < 3; d++)
            if (spc_vals[3 * s + d] == 1) {  // Solver.cs:110-112: "== 1" on a double
                const int32_t dof = node_dof[3 * (int64_t)spc_nodes[s] + d];
                if (dof < 0 || dof >= n_dof) return STAN_HOST_E_ARG;
                red_out[dof] = -1;           // Distinct + Sort + mark (Solver.cs:117-122)
            }
    int32_t reduc = 0;
    for (int64_t i = 0; i < n_dof; i++) {    // Solver.cs:124-132
        if (red_out[i] == -1) reduc++;
        else red_out[i] = reduc;
    }
    if (n_fixed_out) *n_fixed_out = reduc;
    return STAN_HOST_OK;
}

// Solver.cs:136-152
int stan_host_load_vector(int64_t n_dof, const int32_t *node_dof, const int32_t *red,
                          int64_t n_load, const int32_t *load_nodes, const double *load_vals,
                          double *F_out) {
    if (!node_dof || !red || !F_out || (n_load > 0 && (!load_nodes || !load_vals)))
        return STAN_HOST_E_ARG;
    for (int64_t l = 0; l < n_load; l++)
        for (int dir = 0; dir < 3; dir++) {
            const int32_t dof = node_dof[3 * (int64_t)load_nodes[l] + dir];
            if (dof < 0 || dof >= n_dof) return STAN_HOST_E_ARG;
            if (red[dof] != -1) F_out[dof - red[dof]] += load_vals[3 * l + dir];
        }
    return STAN_HOST_OK;
}

// SolverFunctions.cs:520-538 + Solver.cs:171-178
int stan_host_nodal_displacements(int64_t n_nodes, const int32_t *node_dof, const int32_t *red,
                                  const double *U, double *disp_out) {
    if (!node_dof || !red || !U || !disp_out) return STAN_HOST_E_ARG;
    for (int64_t i = 0; i < 3 * n_nodes; i++) {
        const int32_t dof = node_dof[i];
        disp_out[i] = red[dof] == -1 ? 0.0 : U[dof - red[dof]];
    }
    return STAN_HOST_OK;
}

}  // extern "C"
