// internal.h -- shared declarations of libstan_hip.so (not part of the C-ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/stan_hip.h"

#define STAN_SLICE 64      // block rows per ELL slice == wavefront width
#define STAN_MAX_INCIDENT 64  // (element,local node) pairs per node of the FAST symbolic kernel (one lane each); more: k_symbolic_big
#define STAN_MAX_ROW_BLOCKS 96 // slice width (blocks per row) of the FAST numeric kernel's LDS accumulators; wider slices: k_numeric_wide

// HIP call guard: records the error on the context and returns STAN_E_HIP.
#define HIPCHK(ctx, call)                                                              \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);            \
            return (e_ == hipErrorOutOfMemory) ? STAN_E_ALLOC : STAN_E_HIP;            \
        }                                                                              \
    } while (0)

#define STANCHK(call)                 \
    do {                              \
        int rc_ = (call);             \
        if (rc_ != STAN_OK) return rc_; \
    } while (0)

// RCCL entry points, resolved with dlopen at comm_init (no link-time dependency so the
// library loads on a host without RCCL / without a GPU).
struct rccl_api {
    void *handle = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, const void *, int) = nullptr;  // ncclUniqueId by value
    int (*CommDestroy)(void *) = nullptr;
    int (*CommAbort)(void *) = nullptr;   // optional: frees ranks blocked in a collective (multi.hip)
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr;          // optional
    int (*CommCount)(void *, int *) = nullptr;   // optional
    int (*CommUserRank)(void *, int *) = nullptr;
    std::string path;      // the file the entry points live in (dladdr)
    bool reused = false;   // it was already mapped in this process (RTLD_NOLOAD): no second RCCL next to the host's
};

// Device blocks of 8 MB and more are kept by the context when they are freed and handed out again
// to the next request they fit (best fit, at most 1.5x the size asked for): on this stack a
// hipMalloc of tens of GB right after the hipFree of as much takes 0.4-1.8 s, erratically -- at
// 200^3 more than the whole assembly (12 + 32 ms) and 10-25 % of a step (tools/alloc_probe.py).
// Reuse is ordered by the context's stream, so no synchronisation is needed; a failed hipMalloc
// flushes the pool and retries; STAN_OPT_POOL = 0 flushes and disables; destroy frees everything.
struct stan_pool {
    static constexpr size_t MIN_BYTES = 8u << 20;
    static constexpr size_t MAX_BLOCKS = 64;   // parked blocks; one assemble + solve parks ~30
    size_t max_bytes = (size_t)64 << 30;       // parked bytes (STAN_OPT_POOL_MAX_BYTES; init: half the device); oldest go first
    struct blk { void *p; size_t cap; };
    std::vector<blk> avail;
    std::unordered_map<void *, size_t> live;  // pooled-class blocks currently handed out
    bool enabled = true;
    size_t bytes_avail = 0;
    void flush() {
        for (blk &b : avail) hipFree(b.p);
        avail.clear();
        bytes_avail = 0;
    }
};

// events owned by one call: destroyed on every exit path
struct event_bag {
    std::vector<hipEvent_t> ev;
    hipEvent_t make(unsigned flags = hipEventDefault) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, flags) == hipSuccess) ev.push_back(e);
        return e;
    }
    ~event_bag() { for (hipEvent_t e : ev) hipEventDestroy(e); }
};

// Slots of stan_ctx::d_status (device) and ::h_status (pinned host mirror), 64 x int64 each.
// Device and host copies of a slot are not always the same quantity: the host side also parks
// scan totals it copies back (SS_H_*).
enum stan_status_slot {
    SS_ERRBITS = 0,      // device: ERR_* bit mask raised by the assembly kernels
    SS_H_NINC = 1,       // host: total node-element incidences
    SS_H_NHALO = 2,      // host: halo block columns of this rank
    SS_H_NSLOTS = 3,     // host: ELL slots (slot_ptr total)
    SS_H_NBLOCKS = 4,    // host: blocks, followed by
    SS_H_MAXROW = 5,     //       the longest row (copied as a pair from SS_WIDTH_SUM / SS_WIDTH_MAX)
    SS_H_COUNT_A = 6,    // host: scan totals (boundary slices, send rows ...)
    SS_H_COUNT_B = 7,    // host: scan totals (interior slices)
    SS_BAD_ELEM = 8,     // min element index with det J == 0 (LLONG_MAX = none)
    SS_AUX = 9,          // fixed-DOF count (assembly) / first HEX8_G1 element (recovery)
    SS_COUNTER = 10,     // FIXED-48 overflow count; colouring: elements still uncoloured (host)
    SS_H_ERRCOPY = 11,   // host: copy of SS_ERRBITS during the colouring rounds
    SS_MAXDEG = 12,      // device + host: most (element, local node) incidences of one owned row when > STAN_MAX_INCIDENT, else 0
    SS_NGIANT = 13,      // device + host: rows whose symbolic sort does not fit the LDS allotment (global scratch)
    SS_NBIG = 14,        // device + host: rows with more than STAN_MAX_INCIDENT incidences (k_symbolic_big's grid)
    SS_WIDTH_SUM = 16,   // device: sum of row lengths, followed by
    SS_WIDTH_MAX = 17,   //         the longest row (k_slice_width)
    SS_H_CG_STATUS = 16, // host: two 8-word copies of the CG status words (chunk polling)
    SS_H_CG_SCALARS = 40 // host: the CG scalars at the end of a solve
};

// Vectors of the CG, kept by the context between solves (cg.hip: stan_cg_workspace).  They are
// allocated BEFORE the value stream of K is placed by search, and the search's probe multiplies
// with exactly these buffers: whether a block streams fast depends on the PAIR (value block,
// vector block) -- profiles/r02/PLACEMENT.md -- so the pair that is timed is the pair that runs.
struct stan_cg_ws {
    int64_t ng = 0, n3 = 0;        // capacities: gather-sized (owned + pad + halo) / owned-sized, in doubles
    double *xb[2] = {nullptr, nullptr}, *p = nullptr, *r = nullptr;   // ng each
    double *v = nullptr, *w = nullptr, *bh = nullptr, *sv = nullptr;  // n3 each
    double *vw_owner = nullptr;    // non-null: v and w are carved out of this one block (placement.hip, second stage): it is what gets freed
};

// ---- peer-to-peer exchanges of the one-process multi-GPU handle (p2p.hip) ---------------------
// All devices of a group handle live in ONE address space, so the sharded CG needs no RCCL launch
// in its loop: the block that finishes a reduction stores its partial sums into EVERY rank's
// mailbox and then counts itself into every rank's arrival counter; the consumer's stream waits
// for the count (a one-wave polling kernel, or hipStreamWaitValue64: p2p.hip) and the consuming kernel adds the
// partials of all ranks in rank order (the result is the same on every rank, bit for bit, and
// the same as a rank-ordered all-reduce gives).  Halo rows are written straight into the
// neighbour's gather vector, followed by the same kind of count.
constexpr int STAN_P2P_RING = 4;    // mailbox slots / counters in rotation (a rank is never two exchanges ahead)
constexpr int STAN_P2P_MAXR = 16;   // ranks of a group that can exchange peer to peer
struct stan_p2p_dev {               // device-resident, one copy per rank
    int32_t n, me;
    double *mbox[STAN_P2P_MAXR];    // rank q's mailbox [RING][n][4] doubles, fine-grained device memory of q
    unsigned long long *sig_red[STAN_P2P_MAXR][STAN_P2P_RING];   // reduction arrivals at rank q
    unsigned long long *sig_halo[STAN_P2P_MAXR][STAN_P2P_RING];  // halo arrivals at rank q
};
struct stan_p2p {                   // host side, shared by the ranks of a group (owned by multi.hip)
    int n = 0;
    int wait_mode = 1;              // STAN_P2P_WAIT_MODE: 1 (default) a one-wave polling kernel; 0 hipStreamWaitValue64; 2 reductions polled by the consuming kernel itself
    struct rank_res {
        int device = 0;
        double *mbox = nullptr;
        unsigned long long *sig_red[STAN_P2P_RING] = {}, *sig_halo[STAN_P2P_RING] = {};
        stan_p2p_dev *d_dev = nullptr;       // this rank's device copy of the table
        unsigned long long *d_tick = nullptr; // ticket counter of the halo-pack kernel (device)
        int64_t red_calls = 0, halo_calls = 0;   // exchanges so far (identical on every rank)
        unsigned long long red_expect[STAN_P2P_RING] = {}, halo_expect[STAN_P2P_RING] = {};   // arrivals each counter must have reached
        // published at the start of a solve (host barrier behind it): where my neighbours write
        double *vec[5] = {};                 // xb0, xb1, p, r, scale
        int64_t nloc = 0;
        std::vector<int> nbr;
        std::vector<int64_t> recv_off;
    };
    std::vector<rank_res> rk;
    // process-per-GPU form (stan_p2p_ipc_setup): this process owns ONE rank; the other ranks' mailboxes,
    // counters and vectors are device memory of other processes mapped through HIP IPC handles, which the
    // ranks exchange over the RCCL communicator they already have (no change to the host application)
    bool ipc = false;
    int ipc_me = -1;
    void *ipc_block = nullptr;                     // own mailbox + counters: one fine-grained allocation, one handle
    std::vector<void *> ipc_peer_block;            // mapped blocks of the peers
    std::unordered_map<std::string, void *> ipc_open;   // vectors of peers mapped so far, by handle bytes
    std::unordered_set<std::string> ipc_live;           // ... of which the peers published these in the current solve
    // host barrier of the worker threads (abortable)
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    std::atomic<bool> broken{false};
};
int stan_p2p_create(stan_p2p **out, const std::vector<int> &devices, std::string *err);   // called with no worker running
int stan_p2p_rank_setup(stan_p2p *pp, int rank, std::string *err);    // from rank's own thread (its device current)
int stan_p2p_rank_finish(stan_p2p *pp, int rank, std::string *err);   // after every rank's setup: uploads the table
void stan_p2p_rank_release(stan_p2p *pp, int rank);
void stan_p2p_destroy(stan_p2p *pp);
void stan_p2p_abort(stan_p2p *pp);                 // frees every stream wait and every host barrier
void stan_p2p_dump(stan_p2p *pp, FILE *f);         // call counts, expected and actual arrivals of every counter
int stan_p2p_barrier(stan_p2p *pp);                // STAN_E_COMM when aborted / timed out
// Host waits of a peer-to-peer solve are BOUNDED: a stream of this rank that makes no progress for
// stan_p2p_stall_seconds() (a peer died or never published) gets its waits released -- the IPC form releases the
// counters this rank waits on (its own memory), the one-process form every rank's -- the exchange is marked broken
// and the solve returns STAN_E_COMM instead of blocking in hipStreamSynchronize with a spinning queue.
double stan_p2p_stall_seconds();                   // STAN_P2P_STALL_S, default 120
void stan_p2p_release_own(stan_ctx *ctx);          // RELEASE_ALL into this rank's own arrival counters + broken
struct stan_ctx;
int stan_p2p_ipc_setup(stan_ctx *ctx);             // collective over the context's RCCL communicator
void stan_p2p_ipc_release(stan_ctx *ctx);
void stan_p2p_ipc_trim(stan_ctx *ctx);             // end of a solve: close the mappings no peer published this time
struct stan_matrix;
int stan_p2p_reduce_slot(stan_ctx *ctx);           // mailbox slot / counter of this rank's NEXT reduction
int stan_p2p_reduce_wait(stan_ctx *ctx, const unsigned long long **ctr = nullptr, unsigned long long *want = nullptr);   // the wait for it (advances the slot): enqueued, or (wait mode 2, ctr / want given) left to the consuming kernel
const stan_p2p_dev *stan_p2p_table(stan_ctx *ctx);
const double *stan_p2p_mailbox(stan_ctx *ctx, int slot);
int stan_p2p_publish_vectors(stan_ctx *ctx, const stan_matrix *K, double *const vec[5]);
int stan_p2p_halo_exchange(stan_ctx *ctx, stan_matrix *K, double *d_vec);

struct stan_matrix;
struct stan_group;   // multi.hip: one process, several GPUs
struct stan_ctx {
    stan_group *group = nullptr;  // set on the handle stan_hip_init_multi returns: the calls fan out
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    int64_t bad_elem = -1;
    // communicator
    int rank = 0, nranks = 1;
    void *comm = nullptr;      // guarded by comm_mu: the group's host thread may abort it (multi.hip)
    std::mutex comm_mu;
    std::atomic<bool> comm_broken{false};  // the communicator was aborted after a peer failed: no further collectives
    rccl_api nccl;
    stan_p2p *p2p = nullptr;   // set on the rank contexts of a group whose devices can reach each other
    bool comm_p2p = false;     // STAN_OPT_COMM_P2P: the CG's reductions and halo exchanges go peer to peer
    bool result_segment = false;  // group rank: the solve leaves only this rank's own entries of U (no gather)
    // hipFree waits for EVERY stream of the device.  Inside a peer-to-peer solve the other ranks' streams hold
    // waits for exchanges this rank has not issued yet; when ranks share a device (the test topology) a hipFree
    // in the middle of the solve therefore never returns (found as a stall in 7 of 32 runs: the stall dump showed
    // every expectation of the stuck ranks met -- they sat in the hipFree of a temporary).  Blocks released
    // during such a solve are kept and freed when it has ended (stan_cg_device).
    bool defer_frees = false;
    bool peers_share_device = false;   // group rank whose device also carries another rank of the process (multi.hip): frees deferred on every transport
    std::vector<void *> deferred;
    // solver options (include/stan_hip.h STAN_OPT_*)
    bool cg_merit_stop = true;
    int cg_rupdate = 10;
    bool cg_fused_refresh = true;  // A x and A p of a refresh iteration in one matrix pass
    bool cg_single_reduce = false; // Chronopoulos-Gear loop: one reduction point per iteration
    int64_t spmv_small_rows = 150000;   // systems of at most this many block rows: one workgroup per slice (k_spmv_small); 0 = never
    bool cg_defer_x = true;        // merit stop off: x' = x + a p is formed by k_update (one read of p)
    bool cols16 = true;            // SpMV reads the packed column stream where a slice allows it
    int vec_store_nt = 3;          // bit 0: p (k_update), bit 1: r (k_step) leave through non-temporal stores
    bool cg_fold_reduce = true;    // reductions finished by the producing kernel's last block
    bool cg_lazy_scaling = true;   // STAN_OPT_CG_LAZY_SCALING: the first product of the loop brings K into its scaled form (k_spmv_first)
    int cg_refine = 1;             // STAN_OPT_CG_REFINE: reduced-precision streams -- 0 fp64 check only, 1 + refinement passes, 2 + fp64 refresh products
    bool overlap_halo = true;  // interior SpMV on a side stream while the halo is exchanged
    hipStream_t side = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    int assembly_mode = 0;     // 0 row-owner gather (default), 1 element-wave colour scatter
    const void *fold_probe = nullptr;  // placement search: this candidate block is a FOLDED value stream (fold.hip)
    int row_folding = -1;      // STAN_OPT_ROW_FOLDING: -1 = products read the folded streams (fold.hip) when they save > 5 % of the slots; 1 always; 0 never
    int sell_sigma = 1;        // SELL-C-sigma: rows sorted by length inside windows of this many slices (1: inside each slice only)
    int placement_tries = 16;  // > 1: allocate the value stream by search (placement.hip); blocks >= 256 MB only
    int64_t placement_max_bytes = 0;  // bytes of candidates the search may hold at once; 0 = a quarter of the free memory
    float prof_placement_ms_best = 0, prof_placement_ms_worst = 0;
    int prof_placement_candidates = 0;
    int prof_placement_moved_vectors = 0;   // 1: the search ended by re-allocating the CG's vectors
    int prof_colours = 0;
    int spmv_variant = -1; // -1 = auto (launch_spmv picks per value stream); >= 0: A/B lab
    // profiling
    bool profiling = false;
    stan_profile prof{};
    stan_pool pool;
    stan_cg_ws ws;
    std::vector<stan_matrix *> matrices;  // alive matrices of this context (detached when it is destroyed)
    // small pinned host + device scratch for status words
    int64_t *h_status = nullptr;  // pinned, 64 words
    int64_t *d_status = nullptr;  // device, 64 words
};

struct stan_matrix {
    stan_ctx *ctx = nullptr;
    std::vector<stan_matrix *> parts;  // group matrix (multi.hip): the shard of each device
    int64_t n_dof = 0, n_red = 0;
    int64_t nb_glob = 0;  // global block rows
    int64_t r0 = 0, r1 = 0;  // owned block rows
    int64_t u0 = 0, u1 = 0;  // group shard: this rank's entries [u0, u1) of the reduced vectors
    int64_t nloc = 0, nhalo = 0;
    int32_t nslices = 0;
    int64_t nslots = 0;      // total k-slots (each = 64 rows x one block)
    int64_t nblocks = 0;     // structural blocks on this rank
    int32_t max_row_blocks = 0;
    int64_t n_elem_scanned = 0;  // elements this rank holds on the device (sharded host entry: its subset)
    int32_t *d_slot_ptr = nullptr;  // [nslices+1]
    int32_t *d_rowlen = nullptr;    // [nslices*64] blocks per row (by local row)
    int32_t *d_rowof = nullptr;     // [nslices*64] SELL-C-sigma: position in the sliced layout -> local block row
    int32_t *d_posof = nullptr;     // [nslices*64] local block row -> position (slice = pos / 64, lane = pos % 64)
    int sigma = 1;                  // sorting window in slices the matrix was built with
    int32_t *d_cols = nullptr;      // [nslots][64] local block-column index
    uint32_t *d_cols16 = nullptr;   // packed column stream of the SpMV (cg.hip colstream): [pair][64]
    int32_t *d_colbase = nullptr;   //   [nslots] smallest column of each slot
    int32_t *d_pair_ptr = nullptr;  //   [nslices+1]
    uint8_t *d_slice_packed = nullptr;  // [nslices] 1 = slice is in the packed stream
    int64_t slots_packed = 0;       // ELL slots of packed slices (modes 1 and 2 of k_pack_cols) ...
    int64_t slots_packed2 = 0;      // ... of which in two-base slices (4 B more per slot: the second base)
    double *d_vals = nullptr;       // [nslots][9][64]
    float *d_vals32 = nullptr;      // same layout, fp32 copy (mixed precision)
    uint32_t *d_vals48 = nullptr;   // FIXED-48 stream of the scaled values, [slot][14][64] dwords
    bool fx48_refused = false;      // some |a_ij| >= 2 after scaling (K not SPD): fp64 is streamed
    // Folded copy of the streams (fold.hip, STAN_OPT_ROW_FOLDING): long rows lend their tails to the idle slots of
    // the short rows of their slice; padded-slot layout with ~blocks/64 slots per slice.
    int32_t *d_fold_ptr = nullptr;      // [nslices + 1] first folded slot of every slice
    uint32_t *d_fold_meta = nullptr;    // [nslices * 64] own slots (16 bits) | first helper lane << 16 | helper lanes << 24
    int4 *d_fold_plan = nullptr;        // [nslices * 64] own slots, owner lane (-1), offset in the owner's row, slots taken
    int32_t *d_fold_cols = nullptr;     // [nfslots][64]
    uint32_t *d_fold_cols16 = nullptr;  // packed column stream of the folded copy (cg.hip colstream) and its
    int32_t *d_fold_colbase = nullptr;  //   per-slot bases,
    int32_t *d_fold_pair_ptr = nullptr; //   per-slice first pairs,
    uint8_t *d_fold_packed = nullptr;   //   per-slice flags
    int64_t fold_slots_packed = 0;
    double *d_fold_vals = nullptr;      // [nfslots][9][64], built on demand (per value stream)
    float *d_fold_vals32 = nullptr;
    uint32_t *d_fold_vals48 = nullptr;  // [nfslots][14][64]
    int64_t nfslots = 0;
    bool fold_cols_filled = false;
    int fold_state = 0;                 // 0: not examined, 1: planned, -1: not applicable / abandoned (no memory), -2: declined by the auto threshold (> 95 % of the slots)
    int32_t *d_red = nullptr;       // [n_dof] nDOF_reduction
    uint8_t *d_fixmask = nullptr;   // [nb_glob] bit m = DOF m of the node fixed
    double *d_scale = nullptr;      // [3*(nloc+nhalo)] s_i = 1/sqrt(K_ii)
    bool scaled = false;
    // halo plan (all index lists on device, counts on host)
    std::vector<int> nbr;             // neighbour ranks
    std::vector<int64_t> send_off;    // [nbr+1] offsets into d_send_rows
    std::vector<int64_t> recv_off;    // [nbr+1] offsets into the halo region (block cols)
    int32_t *d_send_rows = nullptr;   // local block rows to pack, grouped by neighbour
    int32_t *d_halo_glob = nullptr;   // [nhalo] global block index of each halo column
    double *d_sendbuf = nullptr;      // [3*send_total]
    std::vector<int64_t> row_starts;  // [nranks+1] partition
    // slices whose rows reference no halo column (interior) / at least one (boundary)
    int32_t *d_sl_int = nullptr, *d_sl_bnd = nullptr;
    int32_t n_sl_int = 0, n_sl_bnd = 0;
};

// contiguous block-row partition of the sharded system, cut on slice (64-row) boundaries:
// rank r owns [row_start(r), row_start(r+1)) of the nb block rows (reference DOF order)
static inline int64_t stan_row_start(int64_t nb, int nranks, int r) {
    if (r >= nranks) return nb;
    const int64_t nsl = (nb + 63) / 64;
    const int64_t s = nsl * r / nranks * 64;
    return s > nb ? nb : s;
}

// ---- scan.hip -------------------------------------------------------------------------------
// out[i] = sum_{j<i} in[j] (int32 in, int64 out), out has n+1 entries (out[n] = total).
int stan_scan_exclusive(stan_ctx *ctx, const int32_t *d_in, int64_t *d_out, int64_t n);

// ---- assembly.hip ---------------------------------------------------------------------------
int stan_assemble_device(stan_ctx *ctx, int64_t n_nodes, const double *d_xyz,
                         const int32_t *d_node_dof, int64_t n_elem, const int32_t *d_conn,
                         const int32_t *d_elem_mat, const uint8_t *d_elem_type, int32_t n_mat,
                         const double *mat_E_nu, int64_t n_dof, const int32_t *d_red,
                         stan_matrix **outK);
int stan_ke_batch_device(stan_ctx *ctx, int64_t n, const double *d_xyz8, double E, double nu,
                         const uint8_t *d_type, double *d_out);

// ---- assembly_scatter.hip ---------------------------------------------------------------------
int stan_assemble_colour_scatter(stan_ctx *ctx, stan_matrix *K, int64_t n_elem, const int32_t *d_conn,
                                 const int32_t *d_perm, const double *d_xyz, const int32_t *d_elem_mat,
                                 const uint8_t *d_elem_type, const double *d_lamG, const int64_t *d_ptr,
                                 const int32_t *d_list, long long *d_bad);

// ---- cg.hip ---------------------------------------------------------------------------------
int stan_cg_device(stan_ctx *ctx, stan_matrix *K, const double *d_F, double eps_f,
                   int32_t max_its, int32_t precision_mode, double *d_U, int32_t *term,
                   int32_t *iters, double *rel_res);
int stan_spmv_reduced(stan_ctx *ctx, stan_matrix *K, const double *d_x, double *d_y);
int stan_matrix_diagonal(stan_ctx *ctx, stan_matrix *K, double *d_diag);
int stan_spmv_bench_device(stan_ctx *ctx, stan_matrix *K, int32_t precision_mode, int32_t reps,
                           double *avg_ms);
int stan_stream_bench_device(stan_ctx *ctx, stan_matrix *K, int32_t reps, double *avg_ms, int64_t *bytes);
int stan_spmv_local(stan_ctx *ctx, stan_matrix *K, const double *d_x, double *d_y);
int stan_matrix_make_fp32(stan_ctx *ctx, stan_matrix *K);
// ---- fold.hip -------------------------------------------------------------------------------
int stan_matrix_make_folded(stan_ctx *ctx, stan_matrix *K, int32_t stream_kind, bool plan_only = false);
void stan_matrix_drop_folded_values(stan_ctx *ctx, stan_matrix *K);
void stan_matrix_abandon_folding(stan_ctx *ctx, stan_matrix *K);
int stan_matrix_make_fx48(stan_ctx *ctx, stan_matrix *K);
int stan_matrix_make_cols16(stan_ctx *ctx, stan_matrix *K);
int stan_pack_columns(stan_ctx *ctx, int32_t nslices, int64_t nslots, const int32_t *d_slot_ptr, const int32_t *d_cols,
                      uint32_t **packed_out, int32_t **base_out, int32_t **pair_ptr_out, uint8_t **ok_out, int64_t *slots_packed,
                      int64_t nloc = 0, const int32_t *d_rowof = nullptr, const int32_t *d_rowlen = nullptr, int64_t *slots_packed2 = nullptr);
int stan_matrix_ensure_scaled(stan_ctx *ctx, stan_matrix *K);   // S K S in place (once per matrix)
int stan_cg_workspace(stan_ctx *ctx, const stan_matrix *K);  // (re)allocates ctx->ws for K's sizes
void stan_cg_workspace_free(stan_ctx *ctx);
int stan_cg_workspace_move(stan_ctx *ctx, const stan_matrix *K, bool commit, stan_cg_ws *saved);
size_t stan_cg_products_bytes(const stan_ctx *ctx);   // bytes of a block that holds v and w
void stan_cg_products_set(stan_ctx *ctx, double *block, double *saved[3]);   // v, w carved out of `block` (nullptr: back to saved[]); placement.hip, second stage
void stan_cg_products_adopt(stan_ctx *ctx, double *block, size_t bytes, double *saved[3]);   // keep the carved pair for good, release the saved one

// ---- recovery.hip ---------------------------------------------------------------------------
int stan_recover_device(stan_ctx *ctx, int64_t n_nodes, const double *d_xyz, const double *d_disp,
                        int64_t n_elem, const int32_t *d_conn, const int32_t *d_elem_mat,
                        const uint8_t *d_elem_type, int32_t n_mat, const double *mat_E_nu,
                        double *d_strain, double *d_stress, const int32_t *d_node_dof,
                        double *d_elem_forces, double *d_R);

// ---- comm.cpp -------------------------------------------------------------------------------
int stan_comm_allreduce_sum_f64(stan_ctx *ctx, double *d_buf, size_t count);
int stan_comm_info(stan_ctx *ctx, int *version, int *count, int *rank);
int stan_comm_library(stan_ctx *ctx, std::string *path, int *reused);
int stan_comm_allgather_bytes(stan_ctx *ctx, const void *mine, size_t bytes, void *all);   // host buffers, [nranks*bytes] out
// exchange: pack rows listed in K->d_send_rows from d_vec (3 doubles per block row) and
// receive into d_vec + 3*nloc (halo region).
int stan_comm_halo_exchange(stan_ctx *ctx, stan_matrix *K, double *d_vec);
int stan_comm_allgather_rows(stan_ctx *ctx, stan_matrix *K, double *d_full_blockvec);
void stan_comm_abort(stan_ctx *ctx);

// device memory helpers (see stan_pool)
static inline int stan_dmalloc_bytes(stan_ctx *ctx, void **p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) bytes = 8;
    stan_pool &pool = ctx->pool;
    const bool pooled = pool.enabled && bytes >= stan_pool::MIN_BYTES;
    if (pooled) {
        int best = -1;
        for (int i = 0; i < (int)pool.avail.size(); i++) {
            const size_t cap = pool.avail[i].cap;
            if (cap >= bytes && cap <= bytes + bytes / 2 && (best < 0 || cap < pool.avail[best].cap)) best = i;
        }
        if (best >= 0) {
            *p = pool.avail[best].p;
            pool.live[*p] = pool.avail[best].cap;
            pool.bytes_avail -= pool.avail[best].cap;
            pool.avail.erase(pool.avail.begin() + best);
            return STAN_OK;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess && !pool.avail.empty()) {  // make room and try once more
        (void)hipGetLastError();
        pool.flush();
        e = hipMalloc(p, bytes);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *p = nullptr;
        ctx->err = std::string("hipMalloc(") + std::to_string(bytes) + " B): " + hipGetErrorString(e);
        return STAN_E_ALLOC;
    }
    if (pooled) pool.live[*p] = bytes;
    return STAN_OK;
}
template <typename T>
static inline int stan_dmalloc(stan_ctx *ctx, T **p, size_t count) {
    return stan_dmalloc_bytes(ctx, (void **)p, count * sizeof(T));
}
static inline void stan_dfree(stan_ctx *ctx, void *p) {
    if (!p) return;
    if (ctx) {
        auto it = ctx->pool.live.find(p);
        if (it != ctx->pool.live.end()) {
            const size_t cap = it->second;
            ctx->pool.live.erase(it);
            if (ctx->pool.enabled) {
                ctx->pool.avail.push_back({p, cap});
                ctx->pool.bytes_avail += cap;
                // a host cycling through many sizes, or one sharing the GPU with another
                // allocator: drop the oldest parked blocks beyond the block / byte budget
                while (!ctx->defer_frees && !ctx->pool.avail.empty() &&
                       (ctx->pool.avail.size() > stan_pool::MAX_BLOCKS || ctx->pool.bytes_avail > ctx->pool.max_bytes)) {
                    hipFree(ctx->pool.avail.front().p);
                    ctx->pool.bytes_avail -= ctx->pool.avail.front().cap;
                    ctx->pool.avail.erase(ctx->pool.avail.begin());
                }
                return;
            }
        }
        if (ctx->defer_frees) { ctx->deferred.push_back(p); return; }   // (see stan_ctx::defer_frees)
    }
    hipFree(p);
}
static inline void stan_flush_deferred(stan_ctx *ctx) {
    ctx->defer_frees = false;
    for (void *q : ctx->deferred) hipFree(q);
    ctx->deferred.clear();
}

// ---- multi.hip: fan-out of the public entry points for a group handle ------------------------
void stan_set_global_error(const std::string &msg);   // api.hip: errors raised before a context exists
void stan_group_destroy(stan_ctx *lead);
const char *stan_group_last_error(stan_ctx *lead);
int stan_group_ctx_call(stan_ctx *lead, const std::function<int(stan_ctx *)> &fn);
int stan_group_ctx_call_ranked(stan_ctx *lead, const std::function<int(stan_ctx *, int)> &fn);   // fn(rank context, rank) on every worker
int stan_group_assemble(stan_ctx *lead, int64_t n_nodes, const double *xyz, const int32_t *node_dof,
                        int64_t n_elem, const int32_t *conn, const int32_t *elem_mat,
                        const uint8_t *elem_type, int32_t n_mat, const double *mat_E_nu, int64_t n_dof,
                        const int32_t *red, stan_matrix **outK);
void stan_group_matrix_free(stan_matrix *K);
int stan_group_cg_solve(stan_ctx *lead, stan_matrix *K, const double *F, double eps_f, int32_t max_its,
                        int32_t precision_mode, double *U, int32_t *termination_type, int32_t *iterations,
                        double *rel_residual);
int stan_group_recover(stan_ctx *lead, int64_t n_nodes, const double *xyz, const double *disp, int64_t n_elem,
                       const int32_t *conn, const int32_t *elem_mat, const uint8_t *elem_type, int32_t n_mat,
                       const double *mat_E_nu, double *strain, double *stress);
int stan_group_set_p2p(stan_ctx *lead, bool on);
int stan_group_rank0_call(stan_ctx *lead, const std::function<int(stan_ctx *)> &fn);
stan_ctx *stan_group_rank0(stan_ctx *lead);
stan_ctx *stan_group_rank(stan_ctx *lead, int r);
int stan_group_size(stan_ctx *lead);
// entry points that have no meaning for a group handle (device pointers, single-rank helpers)
#define STAN_NO_GROUP(ctx, what)                                                                   \
    do {                                                                                           \
        if ((ctx) && (ctx)->group) {                                                               \
            (ctx)->err = std::string(what) + ": not available on a multi-device handle";           \
            return STAN_E_UNSUPPORTED;                                                             \
        }                                                                                          \
    } while (0)

// placement.hip
int stan_probe_block(stan_ctx *ctx, const void *p, size_t bytes, float *ms_out);
int stan_dmalloc_streamed(stan_ctx *ctx, void **p, size_t bytes,
                          const std::function<int(const void *, float *, bool)> &probe);
int stan_spmv_probe(stan_ctx *ctx, stan_matrix *K, const void *vals, size_t bytes, int32_t precision,
                    float *ms_out, bool self_pair = false);
