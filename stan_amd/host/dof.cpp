// dof.cpp -- host-side integer steps around the GPU hot path (include/stan_host.h).
#include <algorithm>
#include <atomic>
#include <cstring>
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
    // Database.cs:178-196: first node (NodeLib order) contained in exactly 1, else 2 ... 6 elements
    int64_t first = -1;
    for (int c = 1; c < 7 && first < 0; c++)
        for (int64_t i = 0; i < n_nodes; i++)
            if (eptr[(size_t)i + 1] - eptr[(size_t)i] == c) { first = i; break; }
    if (first < 0) return STAN_HOST_E_NO_START;

    std::vector<uint8_t> pushed((size_t)n_nodes, 0);
    std::vector<int32_t> queue((size_t)n_nodes);
    int64_t head = 0, tail = 0;
    int32_t index = 0;
    auto push_neighbours = [&](int64_t nid) {
        for (int64_t q = eptr[(size_t)nid]; q < eptr[(size_t)nid + 1]; q++) {
            const int32_t *nl = conn + (int64_t)elist[(size_t)q] * 8;
            for (int a = 0; a < 8; a++) {
                const int32_t n = nl[a];
                if (!pushed[(size_t)n]) {
                    pushed[(size_t)n] = 1;
                    queue[(size_t)tail++] = n;
                }
            }
        }
    };
    node_index_out[first] = index++;  // Database.cs:209-211
    pushed[(size_t)first] = 1;
    push_neighbours(first);           // NextNode = Neighbors[FirstNode]
    while (index < n_nodes) {
        if (head >= tail) return STAN_HOST_E_DISCONNECTED;
        const int32_t nid = queue[(size_t)head++];
        node_index_out[nid] = index++;
        push_neighbours(nid);
    }
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
