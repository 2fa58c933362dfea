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
#ifdef __cplusplus
}
#endif
#endif
