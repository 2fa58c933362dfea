// placement_lab.hip -- lab only: where does a slow hipMalloc block lose its time?
// stan_hip_lab_placement_map() allocates `ntries` candidate blocks for the value stream of K,
// copies the values into each, and times (a) the whole SpMV (b) the SpMV restricted to each of
// `nseg` consecutive slice ranges (c) a plain front-to-back read of each segment's bytes.
// ms [ntries * (1 + 2*nseg)]: per candidate {whole, seg SpMV x nseg, seg read x nseg}.
#include <vector>

#include "../internal.h"
#include "stan_hip_lab.h"

int stan_spmv_probe_range(stan_ctx *ctx, stan_matrix *K, const double *vals, int32_t s0, int32_t s1,
                          int reps, float *ms_out, int variant);

extern "C" int stan_hip_lab_placement_map(stan_ctx *ctx, stan_matrix *K, int32_t ntries, int32_t nseg,
                                          int32_t keep_fastest, double *ms, uint64_t *addr) {
    if (!ctx || !K || !ms || ntries < 1 || ntries > 16 || nseg < 1 || nseg > 64 || K->ctx != ctx) return STAN_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)K->nslots * 9 * 64 * 8;
    std::vector<int32_t> sp((size_t)K->nslices + 1);
    HIPCHK(ctx, hipMemcpy(sp.data(), K->d_slot_ptr, sp.size() * 4, hipMemcpyDeviceToHost));
    std::vector<void *> cand;
    const int per = 1 + 2 * nseg;
    for (int t = 0; t < ntries; t++) {
        void *q = nullptr;
        if (t == 0) q = K->d_vals;   // candidate 0 = the block the matrix lives in
        else if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        if (t > 0) HIPCHK(ctx, hipMemcpyAsync(q, K->d_vals, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        cand.push_back(q);
        if (addr) addr[t] = (uint64_t)(uintptr_t)q;
        float f = 0;
        STANCHK(stan_spmv_probe_range(ctx, K, (const double *)q, 0, K->nslices, 10, &f, 9));
        ms[t * per] = f;
        for (int g = 0; g < nseg; g++) {
            const int32_t s0 = (int32_t)((int64_t)K->nslices * g / nseg), s1 = (int32_t)((int64_t)K->nslices * (g + 1) / nseg);
            STANCHK(stan_spmv_probe_range(ctx, K, (const double *)q, s0, s1, 10, &f, 9));
            ms[t * per + 1 + g] = f;
            const size_t b0 = (size_t)sp[s0] * 9 * 64 * 8, b1 = (size_t)sp[s1] * 9 * 64 * 8;
            STANCHK(stan_probe_block(ctx, (const char *)q + b0, b1 - b0, &f));
            ms[t * per + 1 + nseg + g] = f;
        }
    }
    for (int t = (int)cand.size(); t < ntries; t++) ms[t * per] = -1;
    // optionally move the matrix into the fastest candidate (so that a CG can be timed on it)
    size_t best = 0;
    for (size_t t = 1; t < cand.size(); t++) if (ms[t * per] < ms[best * per]) best = t;
    size_t worst = 0;
    for (size_t t = 1; t < cand.size(); t++) if (ms[t * per] > ms[worst * per]) worst = t;
    const size_t pick = keep_fastest == 1 ? best : keep_fastest == 2 ? worst : 0;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t t = 1; t < cand.size(); t++)
        if (t != pick) hipFree(cand[t]);
    if (pick != 0) {   // the old block was handed out by the pool: give it back there
        stan_dfree(ctx, K->d_vals);
        K->d_vals = (double *)cand[pick];
    }
    return STAN_OK;
}

// Whole-SpMV time of each candidate block under several kernel variants (walk order / mapping):
// is a slow block slow for every access order, or only for the lockstep front-to-back walk?
// ms [ntries * nvar].  All candidates are freed afterwards (candidate 0 is K's own block).
extern "C" int stan_hip_lab_placement_variants(stan_ctx *ctx, stan_matrix *K, int32_t ntries, int32_t nvar,
                                               const int32_t *variants, int32_t reps, double *ms) {
    if (!ctx || !K || !ms || !variants || ntries < 1 || ntries > 16 || nvar < 1 || K->ctx != ctx) return STAN_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)K->nslots * 9 * 64 * 8;
    std::vector<void *> cand;
    for (int t = 0; t < ntries; t++) {
        void *q = nullptr;
        if (t == 0) q = K->d_vals;
        else if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        if (t > 0) HIPCHK(ctx, hipMemcpyAsync(q, K->d_vals, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        cand.push_back(q);
    }
    for (int round = 0; round < 2; round++)   // second round overwrites the first: warm
        for (size_t t = 0; t < cand.size(); t++)
            for (int v = 0; v < nvar; v++) {
                float f = 0;
                STANCHK(stan_spmv_probe_range(ctx, K, (const double *)cand[t], 0, K->nslices, reps, &f, variants[v]));
                ms[t * nvar + v] = f;
            }
    for (size_t t = cand.size(); t < (size_t)ntries; t++) for (int v = 0; v < nvar; v++) ms[t * nvar + v] = -1;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t t = 1; t < cand.size(); t++) hipFree(cand[t]);
    return STAN_OK;
}
