/* stan_hip_lab.h -- entry points that exist ONLY in the lab build of the library
 * (build_lab/libstan_hip_lab.so, `make -C stan_amd/csrc lab`, compiled with -DSTAN_LAB).
 * Measurement aids of the A/B runs under tools/; never shipped, never loaded by the product
 * path (stan_amd/hip.py loads it only when STAN_HIP_LIB points at it). */
#ifndef STAN_HIP_LAB_H
#define STAN_HIP_LAB_H
#include "../../../include/stan_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* The same operator converted on the device to scalar CSR (fp64 values, int32 columns: what
 * SURVEY.md section 8d prices and alglib's CRS holds) and multiplied by a CSR-vector kernel,
 * timed like stan_hip_spmv_bench.  max_rel_diff compares its product with the BSELL-64 one.
 * Single-rank contexts, matrix in its current (scaled or not) state. */
int stan_hip_csr_spmv_bench(stan_ctx *ctx, stan_matrix *K, int32_t reps, double *avg_ms,
                            int64_t *bytes_per_launch, double *max_rel_diff);
/* Placement map (lab/placement_lab.hip): ntries candidate blocks for K's value stream; per
 * candidate the whole-SpMV time, the SpMV time of each of nseg consecutive slice ranges and the
 * plain-read time of each range's bytes: ms [ntries * (1 + 2*nseg)], addr [ntries] device addresses.
 * keep_fastest: 0 leave K where it is, 1 move it to the fastest candidate, 2 to the slowest. */
int stan_hip_lab_placement_map(stan_ctx *ctx, stan_matrix *K, int32_t ntries, int32_t nseg,
                               int32_t keep_fastest, double *ms, uint64_t *addr);
/* Whole-SpMV time of each of ntries candidate blocks under each of nvar kernel variants
 * (cg.hip: 0 plain, 9 default, 13 default without the nt hint, 14 odd slices walk backwards,
 * 15 rotated start, 16 hashed start): ms [ntries * nvar]. */
int stan_hip_lab_placement_variants(stan_ctx *ctx, stan_matrix *K, int32_t ntries, int32_t nvar,
                                    const int32_t *variants, int32_t reps, double *ms);
/* What distinguishes a slow block: per candidate (allocation strategy 0 hipMalloc(bytes), 1 next
 * power of two, 2 rounded up to 1 GiB, 3 hipExtMallocWithFlags(uncached), 4 K's own block) the
 * whole-SpMV ms, page-touch ms at 4 KiB / 64 KiB / 2 MiB stride (one load per page of that size:
 * translation cost, hardly any data) and the allocation's wall ms: out [n * 5]. */
int stan_hip_lab_placement_alloc(stan_ctx *ctx, stan_matrix *K, int32_t n, const int32_t *strategy,
                                 double *out, uint64_t *addr);
/* Placement or time?  ntries candidates allocated and filled first, then timed round-robin for
 * nrounds rounds (reps launches per measurement, pause_ms of host sleep between rounds):
 * ms, t_s [nrounds * ntries] (t_s = host seconds since the first measurement; may be NULL). */
int stan_hip_lab_placement_rounds(stan_ctx *ctx, stan_matrix *K, int32_t ntries, int32_t nrounds,
                                  int32_t reps, int32_t pause_ms, double *ms, double *t_s);
/* Does the placement of the vectors matter too?  See lab/placement_lab.hip. */
int stan_hip_lab_placement_cross(stan_ctx *ctx, stan_matrix *K, int32_t ntries, double *out,
                                 double *cross_fast, double *cross_slow, int32_t *i_fast, int32_t *i_slow);
/* The in-CG penalty of the SpMV, isolated: out_ms [15] = product alone back to back / gather vector
 * rewritten before every product / plus a k_step pass over other vectors / only that pass. */
int stan_hip_lab_incg_penalty(stan_ctx *ctx, stan_matrix *K, int32_t reps, double *out_ms);
/* Do other allocators put the vectors into another group?  out [5], see lab/placement_lab.hip. */
int stan_hip_lab_placement_vecalloc(stan_ctx *ctx, stan_matrix *K, double *out);
int stan_hip_lab_placement_vecshape(stan_ctx *ctx, stan_matrix *K, double *out);   /* out [8], see lab/placement_lab.hip */
/* Counter runs for the placement question: reps launches cross-paired, self-paired, cross-paired again
 * (out_ms [3]); see lab/placement_lab.hip and tools/placement_pmc.sh. */
int stan_hip_lab_pairing_pmc(stan_ctx *ctx, stan_matrix *K, int32_t reps, double *out_ms);
#ifdef __cplusplus
}
#endif
#endif
