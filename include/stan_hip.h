/*
 * stan_hip.h -- C-ABI of libstan_hip.so: the MI355X (gfx950) implementation of
 * STAN's linear-static hot path (HEX8 stiffness assembly -> global sparse K ->
 * Jacobi-scaled Conjugate Gradient).
 *
 * The reference (galuszkm/STAN, C#) has no FFI for this path; the boundary is
 * the pair of managed methods SolverLinearStatics calls (Solver.cs:156,162):
 *     alglib.sparsematrix ParallelAssembly_K(Database, int[] nDOF_reduction, int inc, string type)
 *                                                   -- SolverFunctions.cs:117-180
 *     double[] LinearSolver_CG(alglib.sparsematrix K, double[] F, Analysis)
 *                                                   -- SolverFunctions.cs:270-330
 * K is only ever handed from the first to the second, so here it is an opaque,
 * device-resident handle (stan_matrix).  INTEGRATION.md shows the P/Invoke stub.
 *
 * Conventions: blittable types only, caller-allocated buffers, no callbacks.
 * Every function returns 0 on success or a negative STAN_E_* code; the text is
 * available from stan_hip_last_error().  Numerical outcomes of the CG are NOT
 * errors: they are reported in *termination_type with ALGLIB's lincg codes,
 * exactly as the reference prints them (SolverFunctions.cs:308-325) and the
 * solution vector is returned regardless (SolverFunctions.cs:329).
 *
 * Several GPUs, two ways, same sharding underneath (rows of K cut into contiguous
 * block-row ranges in reference DOF order; the CG exchanges halos and reduces its
 * dot products over RCCL):
 *   (a) ONE PROCESS -- what a .NET host or the console driver uses: stan_hip_init_multi
 *       returns a handle that drives n devices (one worker thread and one communicator
 *       rank per device inside the library); every other call is the single-GPU call.
 *   (b) one process per GPU under a launcher (bench.py under torch.distributed.run):
 *       every rank creates an ordinary context, joins a communicator
 *       (stan_hip_comm_init) and makes the same calls with the same full-size arguments.
 */
#ifndef STAN_HIP_H
#define STAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct stan_ctx stan_ctx;
typedef struct stan_matrix stan_matrix;

/* error codes */
#define STAN_OK 0
#define STAN_E_HIP (-1)          /* HIP runtime failure / no device                      */
#define STAN_E_ARG (-2)          /* bad argument                                          */
#define STAN_E_ALLOC (-3)        /* device allocation failed                              */
#define STAN_E_DETJ (-4)         /* det J == 0 at a Gauss point (MatrixST.cs:315-318
                                    throws ArgumentException); element via last_error and
                                    stan_hip_last_bad_element()                           */
#define STAN_E_DOF_LAYOUT (-5)   /* Node.DOF is not {3i,3i+1,3i+2} (Node.cs:218-223)       */
#define STAN_E_VALENCE (-6)      /* (rounds 1-3: more than 64 incidences / 96 coupled nodes at one node.)  The
                                    reference bounds neither (Database.cs:149-176, SolverFunctions.cs:143-173) and since
                                    round 4 neither does the library: high-valence nodes take slow paths of the
                                    symbolic and numeric kernels.  Left: 2^26 incident elements at ONE node
                                    (int32 sort buffer) */
#define STAN_E_COMM (-7)         /* RCCL failure                                           */
#define STAN_E_UNSUPPORTED (-8)  /* element type / precision mode not supported           */

/* element type codes (Element.Type strings, Element.cs:15) */
#define STAN_HEX8_G1 1 /* "HEX8_G1" FE_Library.cs:63-89  */
#define STAN_HEX8_G2 2 /* "HEX8_G2" FE_Library.cs:91-131 */

/* precision_mode of stan_hip_cg_solve */
#define STAN_PREC_FP64 0  /* fp64 matrix, fp64 vectors (reference arithmetic)              */
#define STAN_PREC_MIXED 1 /* fp32 matrix values, fp64 vectors and accumulation             */
/* fp64 arithmetic on a 48-bit fixed-point copy of the Jacobi-scaled matrix: every entry of
 * S K S is in [-1,1] for SPD K, stored as rint(a * 2^46) (absolute error <= 2^-47 = 7e-15 of
 * the unit diagonal); 60 B per 3x3 block instead of 76 B.  If some |a| >= 2 (K not SPD) the
 * fp64 values are streamed instead (stan_profile.value_stream tells). */
#define STAN_PREC_FIXED48 2

/* ---- context ------------------------------------------------------------------------- */
/* `device` = HIP device ordinal this process drives.  Fails loudly (STAN_E_HIP) when no
 * GPU is present: there is no CPU fallback. */
int stan_hip_init(int device, stan_ctx **out);
/* One handle for n_devices GPUs of this node (devices [n_devices] HIP ordinals, NULL = 0..n-1):
 * SURVEY.md section 8b's `stan_hip_init(int n_devices, ...)`.  assemble_hex8, cg_solve,
 * recover_hex8, nodal_forces_hex8, set_option, set_profiling, matrix_info, matrix_free, last_error
 * and destroy take the handle like a single-GPU context; the entry points with DEVICE pointers
 * (*_dev), set_stream, comm_init and the single-rank parity helpers (matrix_to_csr, spmv, ...)
 * return STAN_E_UNSUPPORTED on it.  n_devices == 1 is allowed (no communicator). */
int stan_hip_init_multi(int n_devices, const int *devices, stan_ctx **out);
void stan_hip_destroy(stan_ctx *ctx);
const char *stan_hip_last_error(stan_ctx *ctx);
int64_t stan_hip_last_bad_element(stan_ctx *ctx);
/* Use an existing hipStream_t (e.g. torch's current stream) for all work; NULL = own stream. */
int stan_hip_set_stream(stan_ctx *ctx, void *hip_stream);

/* Solver options (defaults reproduce alglib.lincg as the reference uses it).  The numbers are ABI and stay as they were
 * handed out over the rounds; in numerical order:
 *    1 STAN_OPT_CG_MERIT_STOP        1     alglib's merit-function stop (termination type 7)
 *    2 STAN_OPT_CG_RUPDATE           10    residual recomputed every n iterations (ItsBeforeRUpdate)
 *    3 STAN_OPT_SPMV_VARIANT         -1    SpMV kernel variant (auto)
 *    4 STAN_OPT_OVERLAP_HALO         1     sharded SpMV: interior slices behind the halo exchange
 *    5 STAN_OPT_ASSEMBLY_MODE        0     0 row-owner gather, 1 element wave + colour scatter
 *    6 STAN_OPT_CG_FUSED_REFRESH     1     A x and A p from one matrix pass on refresh iterations
 *    7 STAN_OPT_POOL                 1     freed device blocks stay with the context
 *    8 STAN_OPT_PLACEMENT_TRIES      16    allocation of K's values by search (1 = plain)
 *    9 STAN_OPT_POOL_MAX_BYTES       half the device   byte budget of the parked blocks
 *   10 STAN_OPT_CG_SINGLE_REDUCE     0     Chronopoulos-Gear loop (one reduction point per iteration)
 *   11 STAN_OPT_CG_FOLD_REDUCE       1     reductions finished by the producing kernel's last block
 *   12 STAN_OPT_VEC_STORE_NT         3     non-temporal stores of p (bit 0) and r (bit 1)
 *   13 STAN_OPT_PACKED_COLUMNS       1     16-bit column offsets from per-slot bases
 *   14 STAN_OPT_CG_DEFER_X           1     x' formed next to p' when the merit stop is off
 *   15 STAN_OPT_SPMV_SMALL           1     one workgroup per slice below 150 000 block rows
 *   16 STAN_OPT_PLACEMENT_MAX_BYTES  0     byte budget of the placement search (0 = a quarter of free memory)
 *   17 STAN_OPT_SELL_SIGMA           1     SELL-C-sigma sorting window in slices
 *   18 STAN_OPT_COMM_P2P             0     sharded CG exchanges peer to peer instead of over RCCL
 *   19 STAN_OPT_ROW_FOLDING          -1    folded rows on irregular meshes (auto)
 *   20 STAN_OPT_CG_REFINE            1     reduced-precision streams: 0 fp64 check only, 1 + refinement passes, 2 + fp64 refresh
 *   21 STAN_OPT_CG_LAZY_SCALING      1     the loop's first product brings K into its Jacobi-scaled form (no pass of its own)
 * The descriptions follow in the order the options were added.
 *   STAN_OPT_CG_MERIT_STOP  1 (default): stop with type 7 when the merit function x'Ax-2b'x
 *                           no longer decreases (rounding floor, ~1e-7 relative residual on
 *                           large meshes); 0: iterate until eps_f / max_its only.
 *   STAN_OPT_CG_RUPDATE     residual is recomputed as b-Ax every this many iterations
 *                           (default 10 = ALGLIB's ItsBeforeRUpdate; 0 = never). */
#define STAN_OPT_CG_MERIT_STOP 1
#define STAN_OPT_CG_RUPDATE 2
#define STAN_OPT_OVERLAP_HALO 4 /* 1 (default): sharded SpMV = interior slices on a side stream
                                  while the halo is exchanged, then the boundary slices */
#define STAN_OPT_ASSEMBLY_MODE 5 /* 0 (default): row-owner gather; 1: one element per wavefront +
                                   colour-ordered scatter (the north-star variant, single rank) */
#define STAN_OPT_CG_FUSED_REFRESH 6 /* 1 (default): on refresh iterations A x and A p come from ONE
                                     matrix pass and r = b - (A x + a A p); 0: ALGLIB's literal
                                     second product A (x + a p).  Same value up to rounding. */
#define STAN_OPT_CG_SINGLE_REDUCE 10 /* 0 (default): the classic loop = alglib's recurrences (two reduction points
                           per iteration: p.Ap, then r.r with the merit sum).  1: Chronopoulos-Gear form --
                           the same iterates in exact arithmetic from ONE reduction point per iteration
                           (r.r, r.Ar and the merit sum in one all-reduce of 3 doubles when sharded; two
                           kernels per iteration instead of three).  Same stopping rules and codes;
                           iteration counts may differ by a few (different rounding). */
#define STAN_OPT_CG_FOLD_REDUCE 11 /* 1 (default): the block of a producing kernel that finishes last adds up the
                           per-block partial sums (fixed order); 0: separate one-block reduction launches,
                           same order, same bits. */
#define STAN_OPT_VEC_STORE_NT 12 /* cache policy of the vectors the CG's vector kernels WRITE: bit 0 = the search
                           direction p (k_update; gathered by the next product), bit 1 = the residual r (k_step).
                           Default 3: both leave through non-temporal stores (cg.hip: dirty vector lines left in
                           the cache hierarchy are written back inside the next read-only sweep of the matrix).
                           Same arithmetic for every value. */
#define STAN_OPT_PACKED_COLUMNS 13 /* 1 (default): the SpMV reads block-column indices as 16-bit offsets from a
                           per-slot base, two slots per dword (2 B per block instead of 4; lossless, same
                           products in the same order, same bits); slices whose offsets do not fit 16 bits keep
                           the int32 stream.  0: int32 columns everywhere. */
#define STAN_OPT_CG_DEFER_X 14 /* 1 (default): when STAN_OPT_CG_MERIT_STOP is 0 (nothing needs the new iterate before
                           the direction update) x' = x + alpha p is formed in the kernel that forms
                           p' = r + beta p, so p is read once per iteration for both; same operands, same
                           bits.  With the merit-function stop on (the library default) x' is formed earlier, as
                           alglib does, whatever this option says. */
#define STAN_OPT_SPMV_SMALL 15 /* 1 (default): systems of up to 150 000 block rows (450 k DOF) -- too few slices to fill
                           the chip with one wavefront each -- multiply with one WORKGROUP per slice: four
                           wavefronts take every fourth slot, partial rows added through LDS in a fixed order.
                           Chosen by the global row count (shards agree with the whole matrix); same products,
                           another summation order than the large-system kernel.  0: always the large-system
                           kernel; a value > 1: that many block rows as the limit.  An explicit
                           STAN_OPT_SPMV_VARIANT also selects the large-system kernel. */
#define STAN_OPT_PLACEMENT_TRIES 8 /* 16 (default; bench.py: 32).  1: plain allocation.  n = 2..64: the value array of K is
                           allocated by search (placement.hip) -- the SpMV is ~8 % slower, for the life of the
                           blocks, when the matrix stream and the CG's vectors (allocated first, owned by the
                           context) lie in the same group of device memory, and fresh allocations fall into one
                           group in runs of tens of GB.  Candidates are allocated one after the other; each is
                           timed twice with the SpMV itself: with the context's vectors (the pairing the solve
                           will run) and with vectors carved out of the candidate (by construction the slow,
                           same-group pairing).  The first candidate whose real pairing is 3 % faster than its
                           own reference is kept; one that is not stays allocated while the search goes on (so
                           that the allocator moves on), until n candidates have been timed, the byte budget
                           STAN_OPT_PLACEMENT_MAX_BYTES is reached or free device memory falls under 4 block
                           sizes; then the fastest real pairing is kept -- after one more attempt: the CG's
                           vectors are re-allocated beyond the held candidates and kept if that pairing is
                           clear.  Second stage, when that search ended without a clear pairing (also a block
                           with no rivals within the budget: 400^3): blocks of free / 48 (1-4 GB, same budget)
                           are allocated one after the other, the real pairing timed with the two vectors the
                           products WRITE carved out of each, and the best place kept -- block and all -- if it
                           is 1 % better than the pairing so far and 3 % clear of the reference: the place of
                           the written vector alone decides.  The kept block (up to 4 GB for
                           2 x 8 B x nDOF) stays allocated until the context's vectors are re-sized or it is destroyed.
                           Environment, diagnosis only:
                           STAN_PLACEMENT_TRACE=1 prints every probe on stderr, =sweep walks the blocks
                           whatever the first stage found and keeps nothing.
                           Only blocks of 256 MB and more are searched for; ~10 ms per candidate once
                           per context and size; the block pool keeps the winner; the results do not depend on
                           it.  Destroying a context detaches its matrices: they may be freed afterwards. */
#define STAN_OPT_PLACEMENT_MAX_BYTES 16 /* byte budget of the candidates the placement search holds at the same time
                           (default 0 = a quarter of the device memory that is free when the search starts; the
                           first candidate -- the allocation itself -- is always allowed). */
#define STAN_OPT_SELL_SIGMA 17 /* 1 (default) .. 32: SELL-C-sigma -- inside windows of this many 64-row slices the block rows
                           are sorted by length before they are cut into slices (a slice is as wide as its longest row).
                           Reference-order slices of a regular cube carry 1-2 % padding, of a box with 15 % / 40 % of its
                           elements missing 7 % / 24 %; windows of 32 slices: <= 1.5 % (19 % less device memory on the
                           latter).  The permutation is internal: vectors, CRS export, halo plan and the row partition
                           keep the reference (AssignDOF) order and every row sum keeps its bits.  MEASURED (profiles/
                           r03/SELL_C_SIGMA.md): a wave executes the slots of its slice's longest row, and the time of the
                           product follows that count, not the bytes (STAN_OPT_ROW_FOLDING cuts it); sorted slices have fewer
                           slots but lose the locality of 64 consecutive breadth-first rows: 10-27 % slower with 32,
                           6 % with 4.  Hence the default 1 (rows sorted inside each slice
                           only); larger values trade time for memory.  Applies to the next assembly. */
#define STAN_OPT_ROW_FOLDING 19 /* -1 (default) / 0 / 1: on a mesh whose rows differ in length the SpMV reads a re-packed copy of
                           the streams in which the long rows of a slice lend their tails to the idle slots of its short
                           rows (fold.hip): a wave walks about blocks/64 slots instead of the slots of its slice's
                           longest row, and no row leaves its slice (the gather locality of 64 consecutive AssignDOF rows
                           stays -- SELL-C-sigma windows give that up).  A folded row is summed as own part + pieces:
                           another order than the padded layout's (deterministic; identical for a shard and the whole
                           matrix WITH the option at 1 and STAN_OPT_SELL_SIGMA at 1: in auto mode every shard takes its own
                           > 5 % decision and its own out-of-memory fallback, and sorting windows of sigma > 1 are aligned
                           to the shard's first row, so a row may be folded in a shard and not in the whole matrix -- the
                           results then agree to rounding, not to the bit); rows that are not folded keep their bits.  Built at the first solve next to the
                           padded streams, which the scaling, the export and the placement search keep using.  -1: when
                           the plan saves more than 5 % of the slots (a box with 40 % of its elements missing: 16 %, SpMV
                           -13 ... -19 %, +11 ... +20 % DOF/s; the cube: 0 %, never); 1: always (also for a matrix that the
                           auto rule declined earlier); 0: never. */
#define STAN_OPT_COMM_P2P 18 /* on a one-process multi-device handle (stan_hip_init_multi), or -- process-per-GPU form -- on
                           the context of EVERY rank of a communicator: there the call is a COLLECTIVE (the ranks' HIP IPC
                           handles travel over the communicator; a rank that does not make it leaves the others blocked in
                           an all-gather), the wait kernels need one hardware queue per stream (GPU_MAX_HW_QUEUES >= 2 N + 2
                           when N ranks share a device, as the tests do), and a rank whose stream makes no progress for
                           STAN_P2P_STALL_S seconds (default 120: a peer died) releases its own waits and returns
                           STAN_E_COMM from the solve; the exchange stays refused afterwards -- leave the process, never
                           re-execute it.  0 (default): the sharded CG
                           exchanges over RCCL (2 all-reduce launches + 1 grouped send/recv per iteration).  1: peer to
                           peer -- no collective launch in the loop: the block that finishes a reduction stores this
                           rank's partial sums into every rank's mailbox and counts itself into every rank's arrival
                           counter; the consumer's STREAM waits for the count (a one-wave polling kernel) and the consuming
                           kernel adds the partials in rank order (identical bits on every rank, and the bits of a
                           rank-ordered all-reduce); boundary rows are written straight into the neighbours' gather
                           vectors.  Needs peer access between all devices of the handle (STAN_E_UNSUPPORTED
                           otherwise; also on a handle with more than 16 ranks). */
#define STAN_OPT_CG_REFINE 20 /* STAN_PREC_MIXED / STAN_PREC_FIXED48 only (the fp64 stream is alglib's loop as it is).  The
                           loop of such a solve iterates on a ROUNDED copy of S K S; its own residual recurrence says nothing
                           about K.  Every such solve therefore ends with ONE product on the fp64 values (they stay resident
                           next to their copy): r_t = S F - (S K S) x^, and rel_residual reports ||r_t|| / ||S F||.  Measured
                           before this existed: fp32 copy, 400^3, "type 1 at 9.9e-9" for an answer whose fp64 residual was
                           1.3e-3 (kappa-amplified rounding of the entries).
                           0: check only -- a solve whose recurrence met eps_f while r_t does not returns type 7 (no further
                              progress at this precision), never type 1.
                           1 (default): iterative refinement -- while r_t misses eps_f, another pass of the same loop solves
                              (S K S) d = r_t on the reduced-precision stream to ||r|| <= eps_f ||S F|| and x^ += d in fp64; at
                              most 8 passes, stopped with type 7 when a pass does not halve r_t, type 5 at max_its (counted
                              over all passes); iterations = the sum.
                           2: as 1, and the loop's periodic residual recomputation multiplies with the fp64 values, every 50
                              iterations instead of every STAN_OPT_CG_RUPDATE ("reliable updates": the recurrence is re-anchored
                              to the true residual while the Krylov directions are kept; one pass usually suffices).  Meant for
                              STAN_OPT_CG_MERIT_STOP = 0: the merit sum jumps at a re-anchoring and alglib's rule may read
                              that as "no further progress".
                           MEASURED (148^3, eps 1e-8, merit stop off; profiles/r05/mixed_refine_n148.txt): fp64 1333
                           iterations, 1.56 s.  fp32 copy: refine 0 -- 1333 iterations, 0.96 s, fp64 residual 1.9e-4 (type 7), U
                           off by 1.7e-3; refine 1 -- 3 passes, 2808 iterations, 2.02 s, U within 6e-9 of the oracle's; refine 2 --
                           1 pass, 2346 iterations, 1.70 s, 6e-10.  The fp32 copy halves the bytes per product but an answer of
                           fp64 quality costs MORE time than the fp64 stream: kappa amplifies the 6e-8 rounding of the entries
                           four orders above eps, and a restarted or re-anchored CG pays for it in iterations.  FIXED-48
                           (absolute entry error 7e-15) passes its check at once: 1333 iterations, 1.27 s -- that is the
                           reduced-byte mode that keeps the answer. */
#define STAN_OPT_CG_LAZY_SCALING 21 /* 1 (default): alglib's lincg iterates on S K S; the fp64 loop of a single rank lets its
                           FIRST product multiply every block by s_row s_col on the way, write it back and go on with the scaled
                           block (k_spmv_first) instead of a scaling pass of its own before the first iteration (6.4 GB read +
                           6.4 GB written, 2.7 ms at 148^3).  Same expression: the stored values and every result keep their
                           bits.  0: the separate pass, which every other case keeps anyway (several ranks, the reduced-precision
                           streams, the small-system kernel, a folded copy, an export before the first solve). */
#define STAN_OPT_POOL 7 /* 1 (default): device blocks >= 8 MB freed by the library stay with the
                           context and are reused by its next allocations (a hipMalloc of tens of GB
                           costs 0.4-1.8 s here); 0: release them now, plain hipMalloc/hipFree from then on */
#define STAN_OPT_POOL_MAX_BYTES 9 /* byte budget of the parked blocks (default: half the device memory -- a library inside a
                           foreign host must not sit on the device; the oldest are released beyond it).  Setting it trims
                           at once: 0 releases everything parked now and keeps nothing afterwards, without switching the
                           reuse of live blocks' sizes off.  -1: "this process owns the device" -- nine tenths of the
                           memory that is free at the call (bench.py and stan_solver do; at 400^3 the fp64 values and
                           their fp32 copy are 126 + 63 GB: under the default one of them went back to the driver every
                           step, 4.7 s of hipMalloc per step; with the budget the assembly is 0.24 s, its kernels' time). */
#define STAN_OPT_SPMV_VARIANT 3 /* SpMV kernel variant (cg.hip): -1 = auto (default: 9 for fp64/fp32 streams,
                           12 for FIXED-48), 0 = plain loads + identity mapping, 9 = non-temporal matrix
                           stream + XCD-chunked workgroup mapping, 12 = 9 unrolled by 4: these three add a row's
                           products in the same order (same bits).  20 (round 6) = TWO wavefronts per slice
                           (k_spmv_pair: the slots split at an even slot, the halves added in a fixed order): the
                           same products summed in another order; +9 % at 60^3, +5 % at 80^3, nothing at 72^3 and
                           100^3 (profiles/r06/spmv_pair_kernel_n60_n72_n80_n100.txt), so never chosen by -1.
                           Anything else is STAN_E_ARG. */
int stan_hip_set_option(stan_ctx *ctx, int32_t option, int64_t value);
/* What STAN_OPT_POOL currently keeps: bytes and number of parked device blocks (either may be NULL). */
int stan_hip_pool_info(stan_ctx *ctx, int64_t *bytes_parked, int64_t *blocks_parked);

/* ---- multi-GPU (RCCL over xGMI) -------------------------------------------------------- */
/* Rank 0 creates the 128-byte id, the host distributes it (torch.distributed broadcast,
 * MPI, a file ...), every rank calls comm_init. */
int stan_hip_comm_unique_id(char id[128]);
/* id == NULL makes a DETACHED rank: partition, assembly, plan export and local products
 * work, anything that needs a collective returns STAN_E_COMM (used by the tests to check
 * every rank's shard on one GPU).  nranks == 1 with an id creates a real 1-rank communicator. */
int stan_hip_comm_init(stan_ctx *ctx, int rank, int nranks, const char id[128]);

/* What the sharded CG of this context exchanges over: RCCL's version code (ncclGetVersion; 0 without a
 * communicator library), the communicator's own rank count and rank (ncclCommCount / ncclCommUserRank),
 * p2p = 1 when STAN_OPT_COMM_P2P is in effect.  Any pointer may be NULL. */
int stan_hip_comm_info(stan_ctx *ctx, int32_t *rccl_version, int32_t *comm_ranks, int32_t *comm_rank, int32_t *p2p);
/* The FILE the RCCL entry points were resolved from (realpath of dladdr, NUL-terminated, truncated to capacity;
 * empty before comm_init) and reused = 1 when that library was ALREADY MAPPED in the process and is shared with the
 * host (a PyTorch host brings its own librccl.so): comm_init asks with RTLD_NOLOAD first, so one process never runs
 * two RCCL builds side by side.  path may be NULL with capacity 0; reused may be NULL. */
int stan_hip_comm_library(stan_ctx *ctx, char *path, int64_t capacity, int32_t *reused);

/* ---- assembly: replaces ParallelAssembly_K (SolverFunctions.cs:117-180) ----------------- */
/* xyz            [n_nodes*3]  Node.X/Y/Z in NodeLib (wire) order           Node.cs:12-14
 * node_dof       [n_nodes*3]  Node.DOF after Database.AssignDOF            Database.cs:140-234
 * conn           [n_elem*8]   Element.NList as node *indices* (position in NodeLib order),
 *                             CHEXA order                                  Element.cs:18
 * elem_mat       [n_elem]     index into mat_E_nu (resolved MatLib[MatID]) Element.cs:147
 * elem_type      [n_elem]     STAN_HEX8_G1 / STAN_HEX8_G2                  Element.cs:15
 * mat_E_nu       [n_mat*2]    Material.E, Material.Poisson                 Material.cs:31-56
 * ndof_reduction [n_dof]      -1 = fixed, else #fixed DOFs below           Solver.cs:121-132
 * All pointers are HOST memory, owned by the caller for the duration of the call. */
int stan_hip_assemble_hex8(stan_ctx *ctx, int64_t n_nodes, const double *xyz,
                           const int32_t *node_dof, int64_t n_elem, const int32_t *conn,
                           const int32_t *elem_mat, const uint8_t *elem_type, int32_t n_mat,
                           const double *mat_E_nu, int64_t n_dof,
                           const int32_t *ndof_reduction, stan_matrix **outK);
/* Same, but every array pointer (except mat_E_nu, host) is DEVICE memory already resident
 * in HBM on the context's GPU.  This is the entry bench.py times.  A rank of a sharded run may
 * pass only the elements that touch its block rows, in ascending order of their index in the whole
 * model (stan_host_partition_elements): K's bits do not change, and the element number reported
 * with STAN_E_DETJ is then an index into the arrays that were passed. */
int stan_hip_assemble_hex8_dev(stan_ctx *ctx, int64_t n_nodes, const double *d_xyz,
                               const int32_t *d_node_dof, int64_t n_elem, const int32_t *d_conn,
                               const int32_t *d_elem_mat, const uint8_t *d_elem_type,
                               int32_t n_mat, const double *mat_E_nu, int64_t n_dof,
                               const int32_t *d_ndof_reduction, stan_matrix **outK);
void stan_hip_matrix_free(stan_matrix *K);

/* ---- solve: replaces LinearSolver_CG (SolverFunctions.cs:270-330) ------------------------ */
/* F, U: [N] with N = n_dof - #fixed (the reduced system, as the reference passes them).
 * eps_f / max_its = Analysis.LinSolverTolerance / LinSolverIterMax (Analysis.cs:10-11),
 * both zero -> eps_f = 1e-6 (lincgsetcond).  termination_type: 1 converged, 5 max_its,
 * 7 no further progress (best point returned), -5 not SPD, -4 overflow.
 * rel_residual = ||r||/||b|| of the diagonally scaled system at exit: for STAN_PREC_FP64 the loop's own recurrence
 * (what alglib reports); for STAN_PREC_MIXED / STAN_PREC_FIXED48 the residual of the returned point under the FP64
 * matrix, from one extra product (STAN_OPT_CG_REFINE; stan_profile keeps both).  Type 1 is only ever reported with
 * rel_residual <= eps_f.  Any out pointer except U may be NULL. */
int stan_hip_cg_solve(stan_ctx *ctx, stan_matrix *K, const double *F, double eps_f,
                      int32_t max_its, int32_t precision_mode, double *U,
                      int32_t *termination_type, int32_t *iterations, double *rel_residual);
/* F and U in device memory. */
int stan_hip_cg_solve_dev(stan_ctx *ctx, stan_matrix *K, const double *d_F, double eps_f,
                          int32_t max_its, int32_t precision_mode, double *d_U,
                          int32_t *termination_type, int32_t *iterations, double *rel_residual);

/* ---- stress recovery: replaces Element.Recovery_Stress + Update_StrainStress ------------ */
/* (Element.cs:211-246, 257-267, called from Solver.cs:184-210).  disp [n_nodes*3] = the
 * nodal dU_buffer in NodeLib order (Solver.cs:171-178).  strain/stress [n_elem*48]: per
 * element 8 nodes x {xx,yy,zz,xy,yz,xz} = the 8x6 MatrixST stored in Element.Strain[1] /
 * Stress[1].  A HEX8_G1 element is STAN_E_UNSUPPORTED (the reference throws there). */
int stan_hip_recover_hex8(stan_ctx *ctx, int64_t n_nodes, const double *xyz, const double *disp,
                          int64_t n_elem, const int32_t *conn, const int32_t *elem_mat,
                          const uint8_t *elem_type, int32_t n_mat, const double *mat_E_nu,
                          double *strain, double *stress);
int stan_hip_recover_hex8_dev(stan_ctx *ctx, int64_t n_nodes, const double *d_xyz,
                              const double *d_disp, int64_t n_elem, const int32_t *d_conn,
                              const int32_t *d_elem_mat, const uint8_t *d_elem_type,
                              int32_t n_mat, const double *mat_E_nu, double *d_strain,
                              double *d_stress);

/* The same recovery with the results KEPT ON THE DEVICE(S): what a host that serialises them element by element needs
 * (the console driver's ExportOutput, Solver.cs:454-462: at 148^3 the two arrays are 2.5 GB, and bringing them to the host
 * as whole arrays before the writer starts was a tenth of the run).  stan_hip_results_map copies the rows of elements
 * [e0, e1) into pinned staging memory of the CALLING THREAD and returns pointers to it ([ (e1-e0) * 48 ] each, valid
 * until that thread's next map call): writer threads map the chunk they are encoding, so the download overlaps the
 * encoding and the file writes.  Any number of threads may map concurrently.  On a multi-device handle the elements are
 * cut into one contiguous chunk per device (as stan_hip_recover_hex8 does); a range may span chunks. */
typedef struct stan_results stan_results;
int stan_hip_recover_hex8_keep(stan_ctx *ctx, int64_t n_nodes, const double *xyz, const double *disp,
                               int64_t n_elem, const int32_t *conn, const int32_t *elem_mat,
                               const uint8_t *elem_type, int32_t n_mat, const double *mat_E_nu, stan_results **out);
int stan_hip_results_map(stan_results *res, int64_t e0, int64_t e1, const double **strain, const double **stress);
void stan_hip_results_free(stan_results *res);

/* ---- element nodal forces: replaces Element.Compute_NodalForces + the R assembly ---------- */
/* (Element.cs:248-255, Solver.cs:184-196).  f_e = sum_g BL[g]^T dS[g] det J_g w, where dS[g] is
 * the NODE-extrapolated stress row g that Recovery_Stress left behind -- the reference indexes
 * its node list by Gauss point number and that is kept.  elem_forces [n_elem*24] (node-major,
 * = Element.NodalForces) and/or R [n_dof] (R[DOF] += f, full numbering, before Exclude_BC_DOF);
 * either may be NULL, not both.  The linear-static driver discards R (Solver.cs:199).  R is accumulated with fp64
 * atomics in whatever order the wavefronts arrive: it is the ONE output of this library that is not bit-reproducible
 * from run to run (last-bit differences; the reference's own "+=" under Parallel.ForEach is an unsynchronised race,
 * Solver.cs:194).  elem_forces is deterministic. */
int stan_hip_nodal_forces_hex8(stan_ctx *ctx, int64_t n_nodes, const double *xyz, const double *disp,
                               const int32_t *node_dof, int64_t n_elem, const int32_t *conn,
                               const int32_t *elem_mat, const uint8_t *elem_type, int32_t n_mat,
                               const double *mat_E_nu, int64_t n_dof, double *elem_forces, double *R);

/* ---- introspection / parity helpers ------------------------------------------------------- */
typedef struct stan_matrix_info {
    int64_t n_dof;        /* full DOF count                                      */
    int64_t n_reduced;    /* N                                                   */
    int64_t n_block_rows; /* global 3x3 block rows (= nodes)                     */
    int64_t row_begin;    /* this rank's block-row range [row_begin,row_end)     */
    int64_t row_end;
    int64_t n_halo;       /* halo block columns on this rank                     */
    int64_t n_blocks;     /* structural 3x3 blocks stored on this rank           */
    int64_t n_slots;      /* allocated 64-row ELL slots (incl. padding)          */
    int64_t bytes_matrix; /* device bytes of values + column indices             */
    int32_t scaled;       /* 1 once the CG has applied its diagonal scaling      */
    int32_t max_row_blocks;
    int64_t n_elements_on_device; /* elements this rank uploaded and scanned: all of them on a single
                                     rank; on a rank of a sharded run (host-pointer entry) only those
                                     that touch its rows; summed over the devices of a group handle */
    int32_t sell_sigma;   /* sorting window (slices) the matrix was built with (STAN_OPT_SELL_SIGMA); the padding
                             of the layout is n_slots*64/n_blocks - 1 */
    int32_t folded_slots_permille; /* 0: no folded copy (STAN_OPT_ROW_FOLDING); else 1000 * its slots / n_slots (840: a wave
                                      walks 16 % fewer slots than in the padded layout); known after the first solve */
} stan_matrix_info;
int stan_hip_matrix_info(stan_matrix *K, stan_matrix_info *out);
/* The same for ONE shard of a matrix assembled on a multi-device handle (stan_hip_init_multi): part = rank
 * 0 .. n_devices-1; row_begin / row_end / n_halo / n_blocks are that rank's (the halo block rows it receives per
 * product: SURVEY.md section 8e).  On an ordinary matrix part must be 0 and the call is stan_hip_matrix_info. */
int stan_hip_matrix_part_info(stan_matrix *K, int32_t part, stan_matrix_info *out);

/* K_e of one element (debug/parity; Element.cs:118-155): 24x24 row-major into out[576]. */
int stan_hip_ke_hex8(stan_ctx *ctx, const double xyz8[24], double E, double nu, int32_t type,
                     double out[576]);
/* Batched form: n elements, xyz8 [n*24], type [n], out [n*576] (host pointers). */
int stan_hip_ke_hex8_batch(stan_ctx *ctx, int64_t n, const double *xyz8, double E, double nu,
                           const uint8_t *type, double *out);

/* Export the REDUCED matrix (fixed DOFs removed, indices row - red[row]) as CRS with columns
 * ascending -- what alglib.sparseconverttocrs holds (SolverFunctions.cs:275).
 * upper_only=1: only col >= row (the reference's storage).  Two-call protocol: pass
 * rowptr=col=val=NULL to get *nnz, then call again with buffers [N+1],[nnz],[nnz].
 * The export never changes the matrix.  Before the first solve the values are the assembled bits; a solve
 * keeps K in its scaled form S K S, after which the export divides by s_row s_col on the way out: every
 * value within 1 ulp of the assembled one, the pattern identical.  Single-rank contexts only. */
int stan_hip_matrix_to_csr(stan_ctx *ctx, stan_matrix *K, int32_t upper_only, int64_t *nnz,
                           int64_t *rowptr, int32_t *col, double *val);

/* The shard's halo plan as derived on the device (compare stan_host_partition_plan):
 * row_starts [nranks+1]; halo_glob [n_halo]; nbr [<= nranks]; send_off/recv_off [n_nbr+1];
 * send_rows [send_off[n_nbr]] local rows.  Array pointers may be NULL (sizes only). */
int stan_hip_matrix_plan(stan_ctx *ctx, stan_matrix *K, int64_t *row_starts, int64_t *n_halo,
                         int32_t *halo_glob, int32_t *n_nbr, int32_t *nbr, int64_t *send_off,
                         int32_t *send_rows, int64_t *recv_off);
/* y_owned [3*n_owned] = K_shard x_local, x_local [3*(n_owned + n_halo)] = owned rows then halo
 * columns in plan order (host pointers; the UNSCALED matrix, i.e. before any solve). */
int stan_hip_spmv_local(stan_ctx *ctx, stan_matrix *K, const double *x_local, double *y_owned);

/* diag [N] = K_ii of the reduced system (the unscaled K; alglib's lincg takes its Jacobi scaling s_i = 1/sqrt(K_ii) from
 * it, SolverFunctions.cs:300-305): with it a host can form the scaled residual S (F - K U) the solver reports.  Single-rank
 * contexts only. */
int stan_hip_matrix_diagonal(stan_ctx *ctx, stan_matrix *K, double *diag);
/* y = K x on the reduced system (host x,y of length N); single-rank contexts only. */
int stan_hip_spmv(stan_ctx *ctx, stan_matrix *K, const double *x, double *y);
/* Timing helper for the roofline: `reps` back-to-back launches of the CG's SpMV kernel on
 * the matrix' resident operands, bracketed by HIP events on the context stream.
 * Returns average milliseconds per launch. */
int stan_hip_spmv_bench(stan_ctx *ctx, stan_matrix *K, int32_t precision_mode, int32_t reps,
                        double *avg_ms);

/* The yardstick of that roofline: `reps` read-only sweeps over K's resident fp64 values in the product's own access
 * pattern (one wavefront per slice, XCD-chunked mapping, non-temporal loads) with nothing else -- no columns, no gather,
 * no product vector.  avg_ms per sweep, *bytes = what one sweep reads (n_slots * 64 * 72).  bytes / avg_ms is what
 * THIS device's memory system gives a sweep of THIS block: the product's rate can be read against it
 * (bench.py: roofline.stream_GBs, frac_of_stream).  Single-rank contexts only. */
int stan_hip_stream_bench(stan_ctx *ctx, stan_matrix *K, int32_t reps, double *avg_ms, int64_t *bytes);

/* (The scalar-CSR comparison kernel of round 1, stan_hip_csr_spmv_bench, is a lab aid and lives in
 * the lab build only: stan_amd/csrc/lab/stan_hip_lab.h, `make -C stan_amd/csrc lab`.) */

/* Per-phase device timings of the most recent assemble / solve, measured with HIP events on
 * the context stream (profiling must be enabled first; it adds one event pair per launch). */
typedef struct stan_profile {
    double assemble_ms;      /* symbolic + numeric                                  */
    double symbolic_ms;
    double numeric_ms;
    double cg_ms;            /* whole solve incl. scaling and result gather         */
    double spmv_ms_total;    /* sum over SpMV launches of the last solve            */
    int64_t spmv_launches;
    int64_t spmv_bytes;      /* algorithmic bytes of ONE SpMV launch on this rank   */
    int64_t cg_iteration_vector_bytes; /* vector traffic of one CG iteration        */
    int32_t iterations;
    int32_t termination_type;
    int32_t assembly_colours; /* element colours of the last mode-1 assembly */
    int32_t value_stream;     /* STAN_PREC_* of the stream the last CG actually read */
    double spmv2_ms_total;    /* two-product launches of the refresh iterations (k_spmv2),  */
    int64_t spmv2_launches;   /* NOT included in spmv_ms_total / spmv_launches              */
    int64_t loop_kernel_launches;     /* kernels the CG loop enqueued (incl. the run-ahead)  */
    int64_t loop_collectives;         /* RCCL all-reduces the loop enqueued (halo exchanges not counted) */
    int64_t loop_iterations_enqueued; /* iterations those two counts cover                   */
    int32_t placement_candidates;     /* blocks the last allocation-by-search timed (0: none) */
    float placement_ms_best, placement_ms_worst; /* SpMV probe time of the kept / the slowest candidate */
    int64_t col_slots_packed;         /* ELL slots whose columns the last solve read from the packed stream */
    int32_t placement_moved_vectors;  /* 1: no candidate was clear of the vectors' group and the search re-allocated the CG's vectors instead; 2: only the two vectors the products write (second stage, behind spacer blocks) */
    int32_t repacked_streams;         /* 1: the last solve's products read the folded streams (STAN_OPT_ROW_FOLDING) */
    int64_t loop_stream_waits;        /* peer-to-peer exchanges (STAN_OPT_COMM_P2P): stream waits the loop enqueued instead of collectives */
    double comm_reduce_ms_total;      /* sharded loop: stream time between "reduction issued" and "sums available", summed   */
    int64_t comm_reduce_calls;        /*   over this many reduction points (RCCL all-reduce launches or peer-to-peer waits)   */
    double comm_halo_ms_total;        /* the same for the halo exchanges (pack + send/recv, or pack + peer-to-peer wait)     */
    int64_t comm_halo_calls;
    double rel_residual_recurrence;   /* ||r|| / ||b|| of the loop's own recurrence at exit (all streams) */
    double rel_residual_fp64;         /* reduced-precision streams: ||S F - (S K S) x^|| / ||S F|| with the fp64 values; fp64 stream: -1 */
    int32_t refine_passes;            /* passes of the loop the last solve made (1: no refinement pass was needed) */
    int32_t fp64_products;            /* products with the fp64 values inside a reduced-precision solve (checks, STAN_OPT_CG_REFINE = 2 refreshes) */
    double fp64_products_ms;          /* their stream time, not part of spmv_ms_total */
} stan_profile;
int stan_hip_set_profiling(stan_ctx *ctx, int32_t enabled);
int stan_hip_get_profile(stan_ctx *ctx, stan_profile *out);
/* The profile of ONE rank of a multi-device handle (stan_hip_get_profile reports rank 0's); rank must be 0 on an
 * ordinary context. */
int stan_hip_get_profile_rank(stan_ctx *ctx, int32_t rank, stan_profile *out);
/* Which physical device a rank drives: HIP ordinal and PCI bus id ("0000:c1:00.0"; bus_id [32], either may be NULL).
 * bench.py prints them next to a multi-GPU line. */
int stan_hip_device_info(stan_ctx *ctx, int32_t rank, int32_t *hip_ordinal, char bus_id[32]);

#ifdef __cplusplus
}
#endif
#endif
