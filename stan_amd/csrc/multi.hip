// multi.hip -- ONE host process driving SEVERAL GPUs (stan_hip_init_multi).
//
// The reference is a single process (Solver.cs:18-69; the two hot calls at :156,162), and so is
// any .NET host that P/Invokes this library: it cannot be started once per GPU by a launcher.
// A group context therefore owns one ordinary context per device, each driven by its own worker
// thread (HIP's current device and RCCL's group state are per thread), joined into one RCCL
// communicator.  Every entry point of include/stan_hip.h that takes the group handle fans the
// same call out to the workers -- exactly the calls a process-per-GPU launcher (bench.py under
// torch.distributed.run) makes on its ranks -- and hands back rank 0's results: the sharding,
// the halo exchange, the all-reduces and the result gather are the code of comm.hip / cg.hip,
// unchanged.  The host sees the single-GPU API: init_multi, assemble, cg_solve, matrix_free.
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
        bool has_job = false, done = false, quit = false;
        int rc = 0;
    };
    std::vector<stan_ctx *> ctx;         // one ordinary context per device, rank = index
    std::vector<worker *> w;
    std::string err;
    bool broken = false;                 // a rank failed inside a sharded call: the communicators were aborted
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

// run fn(rank) on every worker, wait for all; first non-zero code wins (its text goes to g->err)
int run_all(stan_group *g, const std::function<int(int)> &fn) {
    const int n = (int)g->w.size();
    for (int r = 0; r < n; r++) {
        stan_group::worker *w = g->w[r];
        std::lock_guard<std::mutex> lk(w->m);
        w->job = [fn, r] { return fn(r); };
        w->has_job = true;
        w->done = false;
        w->cv.notify_all();
    }
    int rc = STAN_OK;
    // A rank that fails inside a sharded call returns while its peers wait for it in a collective
    // (ADVICE r01): once one worker has come back with an error, the others get a grace period and
    // then their communicators are aborted, which makes their queued RCCL work return.
    bool failed = false, aborted = false;
    for (int r = 0; r < n; r++) {
        stan_group::worker *w = g->w[r];
        std::unique_lock<std::mutex> lk(w->m);
        while (!w->done) {
            if (!failed) {
                w->cv.wait_for(lk, std::chrono::milliseconds(200), [&] { return w->done; });
                if (w->done) break;
                lk.unlock();   // has some other rank already failed?
                for (int q = 0; q < n && !failed; q++) {
                    std::lock_guard<std::mutex> lq(g->w[q]->m);
                    failed = g->w[q]->done && g->w[q]->rc != STAN_OK;
                }
                lk.lock();
            } else if (!aborted) {
                if (w->cv.wait_for(lk, std::chrono::seconds(5), [&] { return w->done; })) break;
                lk.unlock();
                for (int q = 0; q < n; q++)
                    if (g->ctx[q]) stan_comm_abort(g->ctx[q]);
                g->broken = aborted = true;
                lk.lock();
            } else
                w->cv.wait(lk, [&] { return w->done; });
        }
        if (w->rc != STAN_OK) failed = true;
        // the code that is reported: the first rank's, except that a communicator error gives way to
        // any other one (after an abort every surviving rank reports STAN_E_COMM: a consequence)
        if (w->rc != STAN_OK && (rc == STAN_OK || (rc == STAN_E_COMM && w->rc != STAN_E_COMM))) {
            rc = w->rc;
            g->err = "rank " + std::to_string(r) + ": " + (g->ctx[r] ? g->ctx[r]->err : std::string("no context"));
        }
    }
    return rc;
}

void stop_workers(stan_group *g) {
    for (stan_group::worker *w : g->w) {
        { std::lock_guard<std::mutex> lk(w->m); w->quit = true; w->cv.notify_all(); }
        if (w->th.joinable()) w->th.join();
        delete w;
    }
    g->w.clear();
}

}  // namespace

extern "C" int stan_hip_init_multi(int n_devices, const int *devices, stan_ctx **out) {
    if (!out || n_devices < 1 || n_devices > 64) return STAN_E_ARG;
    *out = nullptr;
    stan_group *g = new stan_group();
    g->ctx.assign((size_t)n_devices, nullptr);
    for (int r = 0; r < n_devices; r++) {
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
    int rc = run_all(g, [&](int r) {
        const int e = stan_hip_init(devices ? devices[r] : r, &g->ctx[r]);
        if (e) init_err[(size_t)r] = stan_hip_last_error(nullptr);   // thread-local text of THIS worker
        return e;
    });
    char id[128];
    if (rc == STAN_OK && n_devices > 1) rc = stan_hip_comm_unique_id(id);
    if (rc == STAN_OK && n_devices > 1)
        rc = run_all(g, [&](int r) { return stan_hip_comm_init(g->ctx[r], r, n_devices, id); });
    if (rc != STAN_OK) {
        std::string msg = g->err;
        for (size_t r = 0; r < g->ctx.size(); r++)
            if (!init_err[r].empty()) { msg = "rank " + std::to_string(r) + ": " + init_err[r]; break; }
        // contexts that did come up are destroyed by their own threads
        run_all(g, [&](int r) { if (g->ctx[r]) { stan_hip_destroy(g->ctx[r]); g->ctx[r] = nullptr; } return STAN_OK; });
        stop_workers(g);
        delete g;
        delete lead;
        stan_set_global_error("stan_hip_init_multi: " + msg);
        return rc;
    }
    *out = lead;
    return STAN_OK;
}

// ---- the fan-out of each public entry point (called from api.hip when ctx->group is set) ----------

void stan_group_destroy(stan_ctx *lead) {
    stan_group *g = lead->group;
    for (stan_matrix *K : lead->matrices) K->ctx = nullptr;   // group matrices may be freed afterwards
    run_all(g, [&](int r) { stan_hip_destroy(g->ctx[r]); g->ctx[r] = nullptr; return STAN_OK; });
    stop_workers(g);
    delete g;
    delete lead;
}

const char *stan_group_last_error(stan_ctx *lead) {
    return lead->err.empty() ? lead->group->err.c_str() : lead->err.c_str();
}

int stan_group_ctx_call(stan_ctx *lead, const std::function<int(stan_ctx *)> &fn) {
    stan_group *g = lead->group;
    lead->err.clear();
    return run_all(g, [&](int r) { return fn(g->ctx[r]); });
}

int stan_group_assemble(stan_ctx *lead, int64_t n_nodes, const double *xyz, const int32_t *node_dof,
                        int64_t n_elem, const int32_t *conn, const int32_t *elem_mat,
                        const uint8_t *elem_type, int32_t n_mat, const double *mat_E_nu, int64_t n_dof,
                        const int32_t *red, stan_matrix **outK) {
    stan_group *g = lead->group;
    lead->err.clear();
    if (g->broken) { lead->err = "a rank failed earlier and the communicators were aborted: destroy this handle"; return STAN_E_COMM; }
    stan_matrix *K = new stan_matrix();
    K->ctx = lead;
    K->parts.assign(g->ctx.size(), nullptr);
    const int rc = run_all(g, [&](int r) {
        const int e = stan_hip_assemble_hex8(g->ctx[r], n_nodes, xyz, node_dof, n_elem, conn, elem_mat, elem_type,
                                             n_mat, mat_E_nu, n_dof, red, &K->parts[r]);
        if (e == STAN_E_DETJ) lead->bad_elem = g->ctx[r]->bad_elem;
        return e;
    });
    if (rc != STAN_OK) {
        run_all(g, [&](int r) { if (K->parts[r]) stan_hip_matrix_free(K->parts[r]); return STAN_OK; });
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
    lead->matrices.push_back(K);
    *outK = K;
    return STAN_OK;
}

void stan_group_matrix_free(stan_matrix *K) {
    stan_ctx *lead = K->ctx;
    if (lead && lead->group) {
        auto &v = lead->matrices;
        for (size_t i = 0; i < v.size(); i++)
            if (v[i] == K) { v.erase(v.begin() + i); break; }
        run_all(lead->group, [&](int r) { stan_hip_matrix_free(K->parts[r]); return STAN_OK; });
    } else   // the group is gone: its contexts detached the parts when they were destroyed
        for (stan_matrix *p : K->parts) stan_hip_matrix_free(p);
    delete K;
}

int stan_group_cg_solve(stan_ctx *lead, stan_matrix *K, const double *F, double eps_f, int32_t max_its,
                        int32_t precision_mode, double *U, int32_t *termination_type, int32_t *iterations,
                        double *rel_residual) {
    stan_group *g = lead->group;
    lead->err.clear();
    if (g->broken) { lead->err = "a rank failed earlier and the communicators were aborted: destroy this handle"; return STAN_E_COMM; }
    const size_t N = (size_t)K->n_red;
    // every rank ends with the whole U (the result gather of the sharded CG); rank 0 writes the
    // caller's buffer, the others a scratch copy
    std::vector<std::vector<double>> scratch(g->ctx.size());
    // STAN_TEST_FAIL_RANK: test hook only (tests/test_gpu_multi.py: a rank that fails while its peers
    // are inside the solve must not hang the host), like STAN_RCCL_LIB
    const char *inject = getenv("STAN_TEST_FAIL_RANK");
    return run_all(g, [&](int r) {
        if (inject && *inject && atoi(inject) == r) {
            g->ctx[r]->err = "injected failure (STAN_TEST_FAIL_RANK)";
            return (int)STAN_E_HIP;
        }
        double *u = U;
        if (r != 0) { scratch[(size_t)r].resize(N ? N : 1); u = scratch[(size_t)r].data(); }
        return stan_hip_cg_solve(g->ctx[r], K->parts[r], F, eps_f, max_its, precision_mode, u,
                                 r == 0 ? termination_type : nullptr, r == 0 ? iterations : nullptr,
                                 r == 0 ? rel_residual : nullptr);
    });
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
        const int rc = stan_hip_recover_hex8(g->ctx[r], n_nodes, xyz, disp, e1 - e0, conn + 8 * e0, elem_mat + e0,
                                             elem_type + e0, n_mat, mat_E_nu, strain + 48 * e0, stress + 48 * e0);
        if (rc == STAN_E_UNSUPPORTED || rc == STAN_E_DETJ) {   // element numbers of the whole model
            lead->bad_elem = g->ctx[r]->bad_elem + e0;
            g->ctx[r]->err = (rc == STAN_E_DETJ ? "det J == 0 in element " : "stress recovery: HEX8_G1 element ") +
                             std::to_string(lead->bad_elem) +
                             (rc == STAN_E_DETJ ? "" : " (the reference throws: N has one row, Element.cs:242)");
        }
        return rc;
    });
}

stan_ctx *stan_group_rank0(stan_ctx *lead) { return lead->group->ctx[0]; }
int stan_group_size(stan_ctx *lead) { return (int)lead->group->ctx.size(); }
