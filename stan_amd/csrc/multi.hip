// multi.hip -- ONE host process driving SEVERAL GPUs (stan_hip_init_multi).
//
// The reference is a single process (Solver.cs:18-69; the two hot calls at :156,162), and so is
// any .NET host that P/Invokes this library: it cannot be started once per GPU by a launcher.
// A group context therefore owns one ordinary context per device, each driven by its own worker
// thread (HIP's current device and RCCL's group state are per thread), joined into one RCCL
// communicator.  Every entry point of include/stan_hip.h that takes the group handle fans the
// same call out to the workers -- exactly the calls a process-per-GPU launcher (bench.py under
// torch.distributed.run) makes on its ranks: the sharding, the halo exchange and the reductions
// are the code of comm.hip / p2p.hip / cg.hip.  The host sees the single-GPU API: init_multi,
// assemble, cg_solve, matrix_free.
//
// Two things only this form can do, because all ranks share one address space:
//   * the exchanges of the CG loop can go PEER TO PEER (STAN_OPT_COMM_P2P, p2p.hip): no collective
//     launch per iteration;
//   * the result needs no gather: every rank copies ITS entries of U straight into the caller's
//     buffer (stan_matrix::u0/u1), and uploads only its entries of F.
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>

#include "internal.h"

struct stan_group {
    struct worker {
        std::thread th;
        std::mutex m;
        std::condition_variable cv;
        std::function<int()> job;
        bool has_job = false, done = true, quit = false;
        int rc = 0;
    };
    std::vector<stan_ctx *> ctx;         // one ordinary context per device, rank = index
    std::vector<worker *> w;
    std::vector<int> devices;
    std::string err;
    bool broken = false;                 // a rank failed inside a sharded call: the exchanges were aborted
    bool wedged = false;                 // a worker never came back after the abort: its thread is abandoned
    stan_p2p *p2p = nullptr;             // peer-to-peer resources (nullptr: not available, see p2p_why)
    std::string p2p_why;
};

namespace {

void worker_main(stan_group::worker *w) {
    for (;;) {
        std::unique_lock<std::mutex> lk(w->m);
        w->cv.wait(lk, [&] { return w->has_job || w->quit; });
        if (w->quit) return;
        std::function<int()> job = std::move(w->job);
        w->has_job = false;
        lk.unlock();
        const int rc = job();
        lk.lock();
        w->rc = rc;
        w->done = true;
        w->cv.notify_all();
    }
}

// RUN_JOIN: calls without any exchange between the ranks (assembly, stress recovery, options,
// destroy): every worker is simply joined and the first error (by rank) is returned -- a shard that
// fails fast (det J == 0, out of memory) while a peer is still inside a long assembly must not
// cost the handle its communicators.
// RUN_EXCHANGE: calls that can block in an exchange with the other ranks (the solve, the
// communicator's rendezvous): a rank that fails returns while its peers wait for it.  Once one
// worker has come back with an error the others get a grace period; then every communicator is
// aborted (ncclCommAbort from this thread: the one RCCL call that may overtake a blocked one; the
// pointer hand-off is guarded, comm.hip) and the peer-to-peer waits are released; a worker that
// STILL does not return within the final bound is abandoned and the handle reports STAN_E_COMM
// for good -- the host is never left waiting forever.
enum run_mode { RUN_JOIN, RUN_EXCHANGE };
constexpr int GRACE_S = 5, FINAL_S = 60;

int run_all(stan_group *g, const std::function<int(int)> &fn, run_mode mode) {
    const int n = (int)g->w.size();
    if (g->wedged) { g->err = "a worker thread never returned from an aborted call: destroy this handle"; return STAN_E_COMM; }
    for (int r = 0; r < n; r++) {
        stan_group::worker *w = g->w[r];
        std::lock_guard<std::mutex> lk(w->m);
        w->job = [fn, r] { return fn(r); };
        w->has_job = true;
        w->done = false;
        w->cv.notify_all();
    }
    int first_failed = -1;   // the rank that failed FIRST in time is the one worth reporting
    if (mode == RUN_JOIN) {
        for (int r = 0; r < n; r++) {
            stan_group::worker *w = g->w[r];
            std::unique_lock<std::mutex> lk(w->m);
            w->cv.wait(lk, [&] { return w->done; });
            if (w->rc != STAN_OK && first_failed < 0) first_failed = r;
        }
    } else {
        using clock = std::chrono::steady_clock;
        bool aborted = false, dumped = false;
        clock::time_point t_fail, t_abort;
        const clock::time_point t_start = clock::now();
        // STAN_DEBUG_STALL_S = n: a call that has not returned after n seconds writes the state of the
        // peer-to-peer exchanges to stderr once (diagnosis of a stalled multi-GPU solve; the call goes on)
        const char *stall = getenv("STAN_DEBUG_STALL_S");
        for (;;) {
            if (stall && !dumped && g->p2p && clock::now() - t_start > std::chrono::seconds(atoi(stall) > 0 ? atoi(stall) : 30)) {
                stan_p2p_dump(g->p2p, stderr);
                dumped = true;
            }
            int running = 0;
            for (int r = 0; r < n; r++) {
                stan_group::worker *w = g->w[r];
                std::lock_guard<std::mutex> lk(w->m);
                if (!w->done) running++;
                else if (w->rc != STAN_OK && first_failed < 0) { first_failed = r; t_fail = clock::now(); }
            }
            if (running == 0) break;
            if (first_failed >= 0 && !aborted && clock::now() - t_fail > std::chrono::seconds(GRACE_S)) {
                for (int q = 0; q < n; q++)
                    if (g->ctx[q]) stan_comm_abort(g->ctx[q]);
                stan_p2p_abort(g->p2p);
                g->broken = aborted = true;
                t_abort = clock::now();
            }
            if (aborted && clock::now() - t_abort > std::chrono::seconds(FINAL_S)) {
                g->wedged = true;   // (the abandoned jobs hold copies of their arguments, not references to this frame)
                break;
            }
            // sleep until some worker finishes (or 100 ms)
            stan_group::worker *w = nullptr;
            for (int r = 0; r < n && !w; r++) { std::lock_guard<std::mutex> lk(g->w[r]->m); if (!g->w[r]->done) w = g->w[r]; }
            if (w) { std::unique_lock<std::mutex> lk(w->m); w->cv.wait_for(lk, std::chrono::milliseconds(100), [&] { return w->done; }); }
        }
    }
    if (g->wedged) {
        g->err = "rank " + std::to_string(first_failed) + " failed and a peer did not return after the exchanges were aborted";
        return STAN_E_COMM;
    }
    // the code that is reported: the rank that failed first, except that a communicator error gives way to
    // any other one (after an abort every surviving rank reports STAN_E_COMM: a consequence, not the cause)
    int rc = STAN_OK, who = -1;
    if (first_failed >= 0) { rc = g->w[first_failed]->rc; who = first_failed; }
    for (int r = 0; r < n; r++) {
        const int e = g->w[r]->rc;
        if (e != STAN_OK && (rc == STAN_OK || (rc == STAN_E_COMM && e != STAN_E_COMM))) { rc = e; who = r; }
    }
    if (rc != STAN_OK)
        g->err = "rank " + std::to_string(who) + ": " + (g->ctx[who] ? g->ctx[who]->err : std::string("no context"));
    return rc;
}

void stop_workers(stan_group *g) {
    for (stan_group::worker *w : g->w) {
        bool idle;
        { std::lock_guard<std::mutex> lk(w->m); idle = w->done && !w->has_job; w->quit = true; w->cv.notify_all(); }
        if (idle && w->th.joinable()) { w->th.join(); delete w; }
        else if (w->th.joinable()) w->th.detach();   // abandoned inside a call that never returned: leaked on purpose
    }
    g->w.clear();
}

// peer-to-peer resources of the group; failure only means the option is unavailable
void setup_p2p(stan_group *g) {
    const int n = (int)g->devices.size();
    bool shared = false;
    for (int a = 0; a < n; a++)
        for (int b = a + 1; b < n; b++) shared |= g->devices[a] == g->devices[b];
    if (shared) {
        // Ranks that share a device (a test topology) share its hardware queues: the runtime multiplexes
        // streams onto GPU_MAX_HW_QUEUES (default 4) queues, and a stream wait blocks the queue it sits in --
        // with a producer of another rank queued behind it that is a deadlock (measured:
        // profiles/r03/waitvalue_probe_default_hw_queues.txt).  Every stream needs a queue of its own.
        const char *q = getenv("GPU_MAX_HW_QUEUES");
        if (!q || atoi(q) < 2 * n + 2) {
            g->p2p_why = "ranks share a device: peer-to-peer stream waits need GPU_MAX_HW_QUEUES >= " +
                         std::to_string(2 * n + 2) + " in the environment (one hardware queue per stream)";
            return;
        }
    }
    stan_p2p *pp = nullptr;
    if (stan_p2p_create(&pp, g->devices, &g->p2p_why) != STAN_OK) return;
    std::vector<std::string> why((size_t)n);
    int rc = run_all(g, [&](int r) { hipSetDevice(g->devices[r]); return stan_p2p_rank_setup(pp, r, &why[(size_t)r]); }, RUN_JOIN);
    if (rc == STAN_OK)
        rc = run_all(g, [&](int r) { hipSetDevice(g->devices[r]); return stan_p2p_rank_finish(pp, r, &why[(size_t)r]); }, RUN_JOIN);
    if (rc != STAN_OK) {
        for (const std::string &s : why) if (!s.empty()) { g->p2p_why = s; break; }
        run_all(g, [&](int r) { hipSetDevice(g->devices[r]); stan_p2p_rank_release(pp, r); return STAN_OK; }, RUN_JOIN);
        stan_p2p_destroy(pp);
        return;
    }
    g->p2p = pp;
    for (stan_ctx *c : g->ctx) c->p2p = pp;
}

}  // namespace

extern "C" int stan_hip_init_multi(int n_devices, const int *devices, stan_ctx **out) {
    if (!out || n_devices < 1 || n_devices > 64) return STAN_E_ARG;
    *out = nullptr;
    stan_group *g = new stan_group();
    g->ctx.assign((size_t)n_devices, nullptr);
    for (int r = 0; r < n_devices; r++) {
        g->devices.push_back(devices ? devices[r] : r);
        stan_group::worker *w = new stan_group::worker();
        g->w.push_back(w);
        w->th = std::thread(worker_main, w);
    }
    stan_ctx *lead = new stan_ctx();
    lead->group = g;
    lead->nranks = n_devices;
    // phase 1: a context per device (a rank that cannot come up must not leave the others
    // blocked in the communicator's rendezvous); phase 2: every rank joins from its own thread
    std::vector<std::string> init_err((size_t)n_devices);
    int rc = run_all(g, [g, &init_err](int r) {
        const int e = stan_hip_init(g->devices[(size_t)r], &g->ctx[(size_t)r]);
        if (e) init_err[(size_t)r] = stan_hip_last_error(nullptr);   // thread-local text of THIS worker
        return e;
    }, RUN_JOIN);
    // (the rendezvous job captures by value: should a rank hang in it for good, run_all gives up and the
    // abandoned worker must not look at this frame again)
    std::string id(128, '\0');
    if (rc == STAN_OK && n_devices > 1) rc = stan_hip_comm_unique_id(&id[0]);
    if (rc == STAN_OK && n_devices > 1)
        rc = run_all(g, [g, id, n_devices](int r) { return stan_hip_comm_init(g->ctx[(size_t)r], r, n_devices, id.data()); }, RUN_EXCHANGE);
    if (rc != STAN_OK) {
        std::string msg = g->err;
        for (size_t r = 0; r < g->ctx.size(); r++)
            if (!init_err[r].empty()) { msg = "rank " + std::to_string(r) + ": " + init_err[r]; break; }
        // contexts that did come up are destroyed by their own threads
        if (!g->wedged)
            run_all(g, [g](int r) { if (g->ctx[(size_t)r]) { stan_hip_destroy(g->ctx[(size_t)r]); g->ctx[(size_t)r] = nullptr; } return STAN_OK; }, RUN_JOIN);
        stop_workers(g);
        if (!g->wedged) delete g;   // an abandoned worker may still touch the group: leaked with it
        delete lead;
        stan_set_global_error("stan_hip_init_multi: " + msg);
        return rc;
    }
    if (n_devices > 1) {
        // Ranks of this process that SHARE a device (the test topology; a node gives every rank its own): hipFree waits for
        // every stream of the device, i.e. also for the other ranks' queued collectives -- which, over a stream-ordered
        // transport (RCCL; tests/fake_rccl in its asynchronous mode), wait for exchanges THIS rank has not enqueued yet.
        // Such ranks keep their frees for the end of the solve on every transport (stan_ctx::defer_frees; round 6: config 4
        // on 8 ranks of one GPU hung exactly there, rounds 3-5 knew the hazard from the peer-to-peer path only).
        bool shared = false;
        for (int a = 0; a < n_devices; a++)
            for (int b = a + 1; b < n_devices; b++) shared |= g->devices[(size_t)a] == g->devices[(size_t)b];
        for (stan_ctx *c : g->ctx) c->peers_share_device = shared;
        for (stan_ctx *c : g->ctx) c->result_segment = true;
        setup_p2p(g);
    }
    *out = lead;
    return STAN_OK;
}

// ---- the fan-out of each public entry point (called from api.hip when ctx->group is set) ----------

void stan_group_destroy(stan_ctx *lead) {
    stan_group *g = lead->group;
    for (stan_matrix *K : lead->matrices) K->ctx = nullptr;   // group matrices may be freed afterwards
    if (!g->wedged) {
        run_all(g, [g](int r) {
            if (g->p2p) { hipSetDevice(g->devices[(size_t)r]); stan_p2p_rank_release(g->p2p, r); }
            stan_hip_destroy(g->ctx[(size_t)r]);
            g->ctx[(size_t)r] = nullptr;
            return STAN_OK;
        }, RUN_JOIN);
        stop_workers(g);
        if (g->p2p) stan_p2p_destroy(g->p2p);
        delete g;
    } else
        stop_workers(g);   // what an abandoned worker still holds is leaked with the group
    delete lead;
}

const char *stan_group_last_error(stan_ctx *lead) {
    return lead->err.empty() ? lead->group->err.c_str() : lead->err.c_str();
}

int stan_group_ctx_call(stan_ctx *lead, const std::function<int(stan_ctx *)> &fn) {
    stan_group *g = lead->group;
    lead->err.clear();
    return run_all(g, [&](int r) { return fn(g->ctx[(size_t)r]); }, RUN_JOIN);
}

int stan_group_ctx_call_ranked(stan_ctx *lead, const std::function<int(stan_ctx *, int)> &fn) {
    stan_group *g = lead->group;
    lead->err.clear();
    return run_all(g, [&](int r) { return fn(g->ctx[(size_t)r], r); }, RUN_JOIN);
}

// STAN_OPT_COMM_P2P on a group handle
int stan_group_set_p2p(stan_ctx *lead, bool on) {
    stan_group *g = lead->group;
    lead->err.clear();
    if (on && !g->p2p) {
        lead->err = "peer-to-peer exchanges are not available on this handle: " +
                    (g->p2p_why.empty() ? std::string("a single device") : g->p2p_why);
        return STAN_E_UNSUPPORTED;
    }
    for (stan_ctx *c : g->ctx) c->comm_p2p = on;
    return STAN_OK;
}

int stan_group_assemble(stan_ctx *lead, int64_t n_nodes, const double *xyz, const int32_t *node_dof,
                        int64_t n_elem, const int32_t *conn, const int32_t *elem_mat,
                        const uint8_t *elem_type, int32_t n_mat, const double *mat_E_nu, int64_t n_dof,
                        const int32_t *red, stan_matrix **outK) {
    stan_group *g = lead->group;
    lead->err.clear();
    if (g->broken) { lead->err = "a rank failed earlier and the exchanges were aborted: destroy this handle"; return STAN_E_COMM; }
    stan_matrix *K = new stan_matrix();
    K->ctx = lead;
    K->parts.assign(g->ctx.size(), nullptr);
    const int rc = run_all(g, [&](int r) {
        const int e = stan_hip_assemble_hex8(g->ctx[(size_t)r], n_nodes, xyz, node_dof, n_elem, conn, elem_mat, elem_type,
                                             n_mat, mat_E_nu, n_dof, red, &K->parts[(size_t)r]);
        if (e == STAN_E_DETJ) lead->bad_elem = g->ctx[(size_t)r]->bad_elem;
        return e;
    }, RUN_JOIN);
    if (rc != STAN_OK) {
        run_all(g, [&](int r) { if (K->parts[(size_t)r]) stan_hip_matrix_free(K->parts[(size_t)r]); return STAN_OK; }, RUN_JOIN);
        delete K;
        return rc;
    }
    const stan_matrix *p0 = K->parts[0];
    K->n_dof = p0->n_dof; K->n_red = p0->n_red; K->nb_glob = p0->nb_glob;
    K->r0 = 0; K->r1 = p0->nb_glob; K->nloc = p0->nb_glob;
    for (const stan_matrix *p : K->parts) {
        K->nblocks += p->nblocks; K->nslots += p->nslots; K->nhalo += p->nhalo; K->nslices += p->nslices;
        K->n_elem_scanned += p->n_elem_scanned;
        if (p->max_row_blocks > K->max_row_blocks) K->max_row_blocks = p->max_row_blocks;
    }
    // a rank's entries of the reduced vectors are contiguous: [#free DOFs below its first row, ... its last)
    {
        size_t r = 0;
        int64_t nfree = 0;
        for (int64_t d = 0; d <= n_dof; d++) {
            while (r < K->parts.size() && d == 3 * K->parts[r]->r0) { K->parts[r]->u0 = nfree; r++; }
            if (d < n_dof && red[d] != -1) nfree++;
        }
        for (size_t q = 0; q < K->parts.size(); q++)
            K->parts[q]->u1 = q + 1 < K->parts.size() ? K->parts[q + 1]->u0 : nfree;
        K->u0 = 0; K->u1 = nfree;
    }
    lead->matrices.push_back(K);
    *outK = K;
    return STAN_OK;
}

void stan_group_matrix_free(stan_matrix *K) {
    stan_ctx *lead = K->ctx;
    if (lead && lead->group && !lead->group->wedged) {
        auto &v = lead->matrices;
        for (size_t i = 0; i < v.size(); i++)
            if (v[i] == K) { v.erase(v.begin() + i); break; }
        run_all(lead->group, [&](int r) { stan_hip_matrix_free(K->parts[(size_t)r]); return STAN_OK; }, RUN_JOIN);
    } else if (!lead)   // the group is gone: its contexts detached the parts when they were destroyed
        for (stan_matrix *p : K->parts) stan_hip_matrix_free(p);
    delete K;
}

int stan_group_cg_solve(stan_ctx *lead, stan_matrix *K, const double *F, double eps_f, int32_t max_its,
                        int32_t precision_mode, double *U, int32_t *termination_type, int32_t *iterations,
                        double *rel_residual) {
    stan_group *g = lead->group;
    lead->err.clear();
    if (g->broken) { lead->err = "a rank failed earlier and the exchanges were aborted: destroy this handle"; return STAN_E_COMM; }
    // Every rank uploads its entries of F and leaves its entries of U in the caller's buffer (disjoint
    // ranges: stan_hip_cg_solve on a rank whose context has result_segment set).
    // STAN_TEST_FAIL_RANK: honoured only together with the test transport (STAN_RCCL_LIB, tests/fake_rccl):
    // tests/test_gpu_multi.py checks that a rank that fails while its peers are inside the solve does not
    // hang the host.  Without the test transport in the environment the variable is ignored.
    int inject = -1;
    if (const char *t = getenv("STAN_RCCL_LIB"))
        if (*t) if (const char *f = getenv("STAN_TEST_FAIL_RANK")) if (*f) inject = atoi(f);
    // by value: an abandoned worker (see run_all) must not read this frame
    return run_all(g, [=](int r) {
        if (inject == r) {
            g->ctx[(size_t)r]->err = "injected failure (STAN_TEST_FAIL_RANK)";
            return (int)STAN_E_HIP;
        }
        return stan_hip_cg_solve(g->ctx[(size_t)r], K->parts[(size_t)r], F, eps_f, max_its, precision_mode, U,
                                 r == 0 ? termination_type : nullptr, r == 0 ? iterations : nullptr,
                                 r == 0 ? rel_residual : nullptr);
    }, RUN_EXCHANGE);
}

// Stress recovery is per element: the elements are cut into one contiguous chunk per device.
int stan_group_recover(stan_ctx *lead, int64_t n_nodes, const double *xyz, const double *disp, int64_t n_elem,
                       const int32_t *conn, const int32_t *elem_mat, const uint8_t *elem_type, int32_t n_mat,
                       const double *mat_E_nu, double *strain, double *stress) {
    stan_group *g = lead->group;
    lead->err.clear();
    const int64_t n = (int64_t)g->ctx.size();
    return run_all(g, [&](int r) {
        const int64_t e0 = n_elem * r / n, e1 = n_elem * (r + 1) / n;
        if (e1 <= e0) return (int)STAN_OK;
        stan_ctx *c = g->ctx[(size_t)r];
        const int rc = stan_hip_recover_hex8(c, n_nodes, xyz, disp, e1 - e0, conn + 8 * e0, elem_mat + e0,
                                             elem_type + e0, n_mat, mat_E_nu, strain + 48 * e0, stress + 48 * e0);
        if (rc == STAN_E_UNSUPPORTED || rc == STAN_E_DETJ) {   // element numbers of the whole model
            lead->bad_elem = c->bad_elem + e0;
            c->err = (rc == STAN_E_DETJ ? "det J == 0 in element " : "stress recovery: HEX8_G1 element ") +
                     std::to_string(lead->bad_elem) +
                     (rc == STAN_E_DETJ ? "" : " (the reference throws: N has one row, Element.cs:242)");
        }
        return rc;
    }, RUN_JOIN);
}

// single-rank helpers that a group handle redirects to rank 0 (K_e, nodal forces, pool info): run on rank
// 0's worker thread... they are synchronous host calls on its context; the error text follows the handle
int stan_group_rank0_call(stan_ctx *lead, const std::function<int(stan_ctx *)> &fn) {
    stan_group *g = lead->group;
    lead->err.clear();
    const int rc = fn(g->ctx[0]);
    if (rc != STAN_OK) lead->err = "rank 0: " + g->ctx[0]->err;
    return rc;
}

stan_ctx *stan_group_rank0(stan_ctx *lead) { return lead->group->ctx[0]; }
stan_ctx *stan_group_rank(stan_ctx *lead, int r) { return lead->group->ctx[(size_t)r]; }
int stan_group_size(stan_ctx *lead) { return (int)lead->group->ctx.size(); }
