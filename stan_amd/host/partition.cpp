// partition.cpp -- row partition and halo plan of the sharded CG, from the mesh alone.
// The reference has no distributed path (new design).  This is the
// host statement of the plan libstan_hip.so derives on the device (assembly.hip); it exists
// so that the plan can be checked without a GPU (tests/test_distributed.py, gloo) and so
// that the device plan can be compared with it entry by entry (-m gpu tests).
//
// Partition: block rows (nodes in reference DOF order, Node.DOF[0]/3) are cut into nranks
// contiguous ranges on 64-row slice boundaries: start(r) = floor(nslices*r/nranks)*64.
// Halo of rank r: every block row outside its range that shares an element with an owned row.
// Send list r -> q: the owned rows of r that share an element with a row of q (= the rows
// of r that are in q's halo, by structural symmetry).  All lists ascend in global index,
// so both sides agree on the order without any negotiation.
#include <algorithm>
#include <vector>

#include "../../include/stan_host.h"

extern "C" int stan_host_partition_rows(int64_t n_block_rows, int32_t nranks, int64_t *row_starts) {
    if (n_block_rows < 0 || nranks < 1 || !row_starts) return STAN_HOST_E_ARG;
    const int64_t nsl = (n_block_rows + 63) / 64;
    for (int r = 0; r <= nranks; r++) {
        int64_t s = nsl * r / nranks * 64;
        row_starts[r] = s > n_block_rows ? n_block_rows : s;
    }
    row_starts[nranks] = n_block_rows;
    return STAN_HOST_OK;
}

extern "C" int stan_host_partition_plan(int64_t n_nodes, const int32_t *node_index, int64_t n_elem,
                                        const int32_t *conn, int32_t nranks, int32_t rank,
                                        int64_t *row_starts, int64_t *n_halo, int32_t *halo_glob,
                                        int32_t *n_nbr, int32_t *nbr_ranks, int64_t *send_off,
                                        int32_t *send_rows, int64_t *recv_off) {
    if (n_nodes <= 0 || !node_index || n_elem < 0 || (n_elem && !conn) || nranks < 1 || nranks > 64 ||
        rank < 0 || rank >= nranks || !row_starts || !n_halo || !n_nbr)
        return STAN_HOST_E_ARG;
    stan_host_partition_rows(n_nodes, nranks, row_starts);
    const int64_t r0 = row_starts[rank], r1 = row_starts[rank + 1], nloc = r1 - r0;
    auto owner = [&](int64_t g) {
        return (int)(std::upper_bound(row_starts, row_starts + nranks + 1, g) - row_starts) - 1;
    };
    std::vector<uint8_t> is_halo((size_t)n_nodes, 0);
    std::vector<uint64_t> needed_by((size_t)(nloc > 0 ? nloc : 1), 0);  // bit q: rank q needs this row
    for (int64_t e = 0; e < n_elem; e++) {
        int64_t g[8];
        int own[8];
        bool mine = false, other = false;
        for (int a = 0; a < 8; a++) {
            const int32_t nd = conn[e * 8 + a];
            if (nd < 0 || nd >= n_nodes) return STAN_HOST_E_ARG;
            g[a] = node_index[nd];
            if (g[a] < 0 || g[a] >= n_nodes) return STAN_HOST_E_ARG;
            own[a] = owner(g[a]);
            mine |= own[a] == rank;
            other |= own[a] != rank;
        }
        if (!mine || !other) continue;
        for (int a = 0; a < 8; a++)
            if (own[a] != rank) {
                is_halo[(size_t)g[a]] = 1;
                for (int b = 0; b < 8; b++)
                    if (own[b] == rank) needed_by[(size_t)(g[b] - r0)] |= 1ull << own[a];
            }
    }
    int64_t nh = 0;
    for (int64_t g = 0; g < n_nodes; g++)
        if (is_halo[(size_t)g]) {
            if (halo_glob) halo_glob[nh] = (int32_t)g;
            nh++;
        }
    *n_halo = nh;
    int nn = 0;
    int64_t soff = 0, roff = 0;
    if (send_off) send_off[0] = 0;
    if (recv_off) recv_off[0] = 0;
    for (int q = 0; q < nranks; q++) {
        if (q == rank) continue;
        int64_t nrecv = 0;
        for (int64_t g = row_starts[q]; g < row_starts[q + 1]; g++) nrecv += is_halo[(size_t)g];
        if (nrecv == 0) continue;
        for (int64_t i = 0; i < nloc; i++)
            if (needed_by[(size_t)i] >> q & 1) {
                if (send_rows) send_rows[soff] = (int32_t)i;
                soff++;
            }
        roff += nrecv;
        if (nbr_ranks) nbr_ranks[nn] = q;
        nn++;
        if (send_off) send_off[nn] = soff;
        if (recv_off) recv_off[nn] = roff;
    }
    *n_nbr = nn;
    return STAN_HOST_OK;
}

// Elements a rank has to hold: those with at least one node in its block-row range (boundary
// elements belong to both owners -- every rank assembles its rows without communication).
// elem_idx_out [<= n_elem] ascending element indices; *n_out their number.  The host-pointer entry
// of libstan_hip.so does this filtering itself; a launcher that keeps its inputs resident on the
// device (stan_hip_assemble_hex8_dev) uses this to upload only the subset.
extern "C" int stan_host_partition_elements(int64_t n_nodes, const int32_t *node_index, int64_t n_elem,
                                            const int32_t *conn, int32_t nranks, int32_t rank,
                                            int32_t *elem_idx_out, int64_t *n_out) {
    if (n_nodes <= 0 || !node_index || n_elem < 0 || (n_elem && !conn) || nranks < 1 || rank < 0 ||
        rank >= nranks || !n_out || (n_elem && !elem_idx_out))
        return STAN_HOST_E_ARG;
    std::vector<int64_t> rs((size_t)nranks + 1);
    stan_host_partition_rows(n_nodes, nranks, rs.data());
    const int64_t r0 = rs[(size_t)rank], r1 = rs[(size_t)rank + 1];
    int64_t n = 0;
    for (int64_t e = 0; e < n_elem; e++) {
        bool mine = false;
        for (int a = 0; a < 8; a++) {
            const int32_t nd = conn[e * 8 + a];
            if (nd < 0 || nd >= n_nodes) return STAN_HOST_E_ARG;
            const int64_t row = node_index[nd];
            mine |= row >= r0 && row < r1;
        }
        if (mine) elem_idx_out[n++] = (int32_t)e;
    }
    *n_out = n;
    return STAN_HOST_OK;
}
