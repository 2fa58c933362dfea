// placement_lab.hip -- lab only: where does a slow hipMalloc block lose its time?
// stan_hip_lab_placement_map() allocates `ntries` candidate blocks for the value stream of K,
// copies the values into each, and times (a) the whole SpMV (b) the SpMV restricted to each of
// `nseg` consecutive slice ranges (c) a plain front-to-back read of each segment's bytes.
// ms [ntries * (1 + 2*nseg)]: per candidate {whole, seg SpMV x nseg, seg read x nseg}.
#include <chrono>
#include <thread>
#include <vector>

#include "../internal.h"
#include "stan_hip_lab.h"

int stan_spmv_probe_range(stan_ctx *ctx, stan_matrix *K, const double *vals, int32_t s0, int32_t s1,
                          int reps, float *ms_out, int variant, double *xy_region = nullptr, double *y_region = nullptr);

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

// ---- what distinguishes a slow block?  page-touch probes + allocation strategies ------------------
namespace {
// one 8-B load per `stride` bytes: little data, one translation per page of that size
__global__ void k_touch(const char *p, size_t bytes, size_t stride, size_t phase, double *sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t off = i * stride + phase;
    if (off + 8 > bytes) return;
    const double v = __builtin_nontemporal_load((const double *)(p + off));
    if (v == 0.1234567890123) sink[0] = v;
}
int touch_ms(stan_ctx *ctx, const void *p, size_t bytes, size_t stride, float *ms) {
    const size_t n = bytes / stride;
    const unsigned grid = (unsigned)((n + 255) / 256);
    event_bag ev;
    hipEvent_t a = ev.make(), b = ev.make();
    const int reps = stride >= (1u << 20) ? 64 : 8;
    for (int r = 0; r < reps + 1; r++) {
        if (r == 1) hipEventRecord(a, ctx->stream);
        hipLaunchKernelGGL(k_touch, dim3(grid ? grid : 1), dim3(256), 0, ctx->stream, (const char *)p, bytes, stride,
                           (size_t)(r * 4096 % (stride > 4096 ? stride : 4096)), (double *)(ctx->d_status + SS_AUX));
    }
    hipEventRecord(b, ctx->stream);
    HIPCHK(ctx, hipEventSynchronize(b));
    hipEventElapsedTime(ms, a, b);
    *ms /= reps;
    return STAN_OK;
}
}  // namespace

// strategy per candidate: 0 hipMalloc(bytes), 1 hipMalloc(next power of two), 2 hipMalloc(bytes
// rounded up to 1 GiB), 3 hipExtMallocWithFlags(hipDeviceMallocContiguous? -> see code), 4 = K's own block.
// out [n * 5]: whole-SpMV ms, touch ms at 4 KiB / 64 KiB / 2 MiB stride, allocation ms.
extern "C" int stan_hip_lab_placement_alloc(stan_ctx *ctx, stan_matrix *K, int32_t n, const int32_t *strategy,
                                            double *out, uint64_t *addr) {
    if (!ctx || !K || !out || !strategy || n < 1 || n > 32 || K->ctx != ctx) return STAN_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)K->nslots * 9 * 64 * 8;
    std::vector<void *> cand((size_t)n, nullptr);
    for (int t = 0; t < n; t++) {
        size_t req = bytes;
        if (strategy[t] == 1) { req = 1; while (req < bytes) req <<= 1; }
        if (strategy[t] == 2) req = (bytes + ((size_t)1 << 30) - 1) >> 30 << 30;
        void *q = nullptr;
        event_bag ev;
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipSuccess;
        if (strategy[t] == 4) q = K->d_vals;
        else if (strategy[t] == 3) e = hipExtMallocWithFlags(&q, req, hipDeviceMallocUncached);
        else e = hipMalloc(&q, req);
        const double alloc_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (e != hipSuccess) { (void)hipGetLastError(); for (int v = 0; v < 5; v++) out[t * 5 + v] = -1; continue; }
        cand[(size_t)t] = q;
        if (addr) addr[t] = (uint64_t)(uintptr_t)q;
        if (q != K->d_vals) HIPCHK(ctx, hipMemcpyAsync(q, K->d_vals, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        float f = 0;
        STANCHK(stan_spmv_probe_range(ctx, K, (const double *)q, 0, K->nslices, 10, &f, 9));
        out[t * 5] = f;
        STANCHK(touch_ms(ctx, q, bytes, 4096, &f)); out[t * 5 + 1] = f;
        STANCHK(touch_ms(ctx, q, bytes, 65536, &f)); out[t * 5 + 2] = f;
        STANCHK(touch_ms(ctx, q, bytes, (size_t)2 << 20, &f)); out[t * 5 + 3] = f;
        out[t * 5 + 4] = alloc_ms;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int t = 0; t < n; t++)
        if (cand[(size_t)t] && cand[(size_t)t] != K->d_vals) hipFree(cand[(size_t)t]);
    return STAN_OK;
}

// Placement or time?  All candidates are allocated and filled FIRST, then timed round-robin:
// ms [nrounds * ntries] (whole-SpMV, `reps` launches each), t_s [nrounds * ntries] host seconds
// since the first measurement.  A block that is slow in every round is slow by placement; rounds
// in which every block is slow are the device's state at that time.
extern "C" int stan_hip_lab_placement_rounds(stan_ctx *ctx, stan_matrix *K, int32_t ntries, int32_t nrounds,
                                             int32_t reps, int32_t pause_ms, double *ms, double *t_s) {
    if (!ctx || !K || !ms || ntries < 1 || ntries > 16 || nrounds < 1 || K->ctx != ctx) return STAN_E_ARG;
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
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < nrounds; r++) {
        for (size_t t = 0; t < (size_t)ntries; t++) {
            float f = -1;
            if (t < cand.size()) STANCHK(stan_spmv_probe_range(ctx, K, (const double *)cand[t], 0, K->nslices, reps, &f, 9));
            ms[(size_t)r * ntries + t] = f;
            if (t_s) t_s[(size_t)r * ntries + t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        if (pause_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(pause_ms));
    }
    for (size_t t = 1; t < cand.size(); t++) hipFree(cand[t]);
    return STAN_OK;
}

// Does the placement of the VECTORS matter too?  ntries blocks are classified with the value stream
// in each (vectors from the pool); then the value stream stays in the fastest (and in the slowest)
// block while the gather vector x and the product y are carved out of each OTHER block in turn.
// out [ntries]: class probe per block; cross_fast / cross_slow [ntries]: SpMV ms with the values in
// the fastest / slowest block and x, y inside block t (-1 for t == that block).
extern "C" int stan_hip_lab_placement_cross(stan_ctx *ctx, stan_matrix *K, int32_t ntries, double *out,
                                            double *cross_fast, double *cross_slow, int32_t *i_fast, int32_t *i_slow) {
    if (!ctx || !K || !out || !cross_fast || !cross_slow || ntries < 2 || ntries > 32 || K->ctx != ctx) return STAN_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)K->nslots * 9 * 64 * 8;
    std::vector<void *> cand;
    for (int t = 0; t < ntries; t++) {
        void *q = nullptr;
        if (t == 0) q = K->d_vals;
        else if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        if (t > 0) HIPCHK(ctx, hipMemcpyAsync(q, K->d_vals, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        cand.push_back(q);
        float f = 0;
        STANCHK(stan_spmv_probe_range(ctx, K, (const double *)q, 0, K->nslices, 10, &f, 9));
        out[t] = f;
    }
    const int n = (int)cand.size();
    for (int t = n; t < ntries; t++) out[t] = cross_fast[t] = cross_slow[t] = -1;
    int bf = 0, bs = 0;
    for (int t = 1; t < n; t++) { if (out[t] < out[bf]) bf = t; if (out[t] > out[bs]) bs = t; }
    if (i_fast) *i_fast = bf;
    if (i_slow) *i_slow = bs;
    // vectors inside block t: block t's matrix copy is overwritten at its front (only blocks other
    // than the one the values are streamed from)
    for (int pass = 0; pass < 2; pass++) {
        const int src = pass == 0 ? bf : bs;
        double *res = pass == 0 ? cross_fast : cross_slow;
        for (int t = 0; t < n; t++) {
            if (t == 0 && src != 0) { res[t] = -1; continue; }   // block 0 is K's own: keep it intact
            if (t == 0 && src == 0) { res[t] = -1; continue; }
            // t == src: the vectors sit inside the very block the values stream from (front of it; the
            // values there are overwritten, which does not matter for a timing)
            float f = 0;
            STANCHK(stan_spmv_probe_range(ctx, K, (const double *)cand[src], 0, K->nslices, 10, &f, 9, (double *)cand[t]));
            res[t] = f;
        }
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int t = 1; t < n; t++) hipFree(cand[t]);
    return STAN_OK;
}

int stan_spmv_incg_lab(stan_ctx *ctx, stan_matrix *K, int reps, double *out_ms);
extern "C" int stan_hip_lab_incg_penalty(stan_ctx *ctx, stan_matrix *K, int32_t reps, double *out_ms) {
    if (!ctx || !K || !out_ms || reps < 1 || K->ctx != ctx) return STAN_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return stan_spmv_incg_lab(ctx, K, reps, out_ms);
}

// Do other allocators put the VECTORS into another group than hipMalloc puts the values?
// out [5]: SpMV ms with the values in K's own block and x, y (0) inside that block (the same-group
// reference), (1) in a fresh hipMalloc block, (2) from hipMallocAsync (stream-ordered pool),
// (3) in a virtual-memory-API mapping (hipMemCreate / hipMemMap), (4) the context's vectors.
extern "C" int stan_hip_lab_placement_vecalloc(stan_ctx *ctx, stan_matrix *K, double *out) {
    if (!ctx || !K || !out || K->ctx != ctx) return STAN_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t vbytes = (size_t)1 << 30;
    for (int k = 0; k < 5; k++) out[k] = -1;
    float f = 0;
    // keep a copy of the front of the values (mode 0 overwrites it)
    void *save = nullptr;
    HIPCHK(ctx, hipMalloc(&save, vbytes));
    HIPCHK(ctx, hipMemcpy(save, K->d_vals, vbytes, hipMemcpyDeviceToDevice));
    STANCHK(stan_spmv_probe_range(ctx, K, K->d_vals, 0, K->nslices, 10, &f, 9, K->d_vals)); out[0] = f;
    HIPCHK(ctx, hipMemcpy(K->d_vals, save, vbytes, hipMemcpyDeviceToDevice));
    hipFree(save);
    void *a = nullptr;
    if (hipMalloc(&a, vbytes) == hipSuccess) {
        STANCHK(stan_spmv_probe_range(ctx, K, K->d_vals, 0, K->nslices, 10, &f, 9, (double *)a)); out[1] = f;
        hipFree(a);
    }
    a = nullptr;
    if (hipMallocAsync(&a, vbytes, ctx->stream) == hipSuccess) {
        STANCHK(stan_spmv_probe_range(ctx, K, K->d_vals, 0, K->nslices, 10, &f, 9, (double *)a)); out[2] = f;
        hipFreeAsync(a, ctx->stream);
        hipStreamSynchronize(ctx->stream);
    } else (void)hipGetLastError();
    {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = ctx->device;
        size_t gran = 0;
        hipMemGenericAllocationHandle_t h;
        hipDeviceptr_t va = nullptr;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && gran > 0) {
            const size_t sz = (vbytes + gran - 1) / gran * gran;
            if (hipMemCreate(&h, sz, &prop, 0) == hipSuccess) {
                if (hipMemAddressReserve(&va, sz, 0, nullptr, 0) == hipSuccess) {
                    hipMemAccessDesc acc = {};
                    acc.location = prop.location;
                    acc.flags = hipMemAccessFlagsProtReadWrite;
                    if (hipMemMap(va, sz, 0, h, 0) == hipSuccess && hipMemSetAccess(va, sz, &acc, 1) == hipSuccess) {
                        STANCHK(stan_spmv_probe_range(ctx, K, K->d_vals, 0, K->nslices, 10, &f, 9, (double *)va)); out[3] = f;
                        hipMemUnmap(va, sz);
                    }
                    hipMemAddressFree(va, sz);
                }
                hipMemRelease(h);
            }
        }
        (void)hipGetLastError();
    }
    STANCHK(stan_cg_workspace(ctx, K));
    {   // the context's own vectors: region = ws.p .. (x) and ws.v (y) are separate blocks; use the probe
        float t = 0;
        STANCHK(stan_spmv_probe(ctx, K, K->d_vals, (size_t)K->nslots * 9 * 64 * 8, STAN_PREC_FP64, &t, false)); out[4] = t;
    }
    return STAN_OK;
}

// One block or several, big or small, earlier or later?  Values in K's own block (plain allocation);
// out [8]: x, y in (0) two fresh 79 MB-class blocks (ng doubles each), (1) one fresh block of 2 ng,
// (2) one fresh 1 GiB block, (3) blocks 3 and 5 of eight fresh ng-sized blocks (what stan_cg_workspace
// allocates), (4)-(7) the same four again in the opposite order of allocation.
extern "C" int stan_hip_lab_placement_vecshape(stan_ctx *ctx, stan_matrix *K, double *out) {
    if (!ctx || !K || !out || K->ctx != ctx) return STAN_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t npad = (int64_t)K->nslices * 64;
    const size_t ng = (size_t)(3 * ((npad > K->nloc + K->nhalo ? npad : K->nloc + K->nhalo)));
    const size_t vb = (ng * 8 + 4095) & ~(size_t)4095;
    for (int k = 0; k < 8; k++) out[k] = -1;
    float f = 0;
    for (int pass = 0; pass < 2; pass++) {
        for (int kk = 0; kk < 4; kk++) {
            const int kind = pass == 0 ? kk : 3 - kk;
            std::vector<void *> blk;
            double *x = nullptr, *y = nullptr;
            auto get = [&](size_t b) { void *q = nullptr; if (hipMalloc(&q, b) != hipSuccess) { (void)hipGetLastError(); q = nullptr; } if (q) blk.push_back(q); return (double *)q; };
            if (kind == 0) { x = get(vb); y = get(vb); }
            if (kind == 1) { x = get(2 * vb); y = x ? x + vb / 8 : nullptr; }
            if (kind == 2) { x = get((size_t)1 << 30); y = x ? x + vb / 8 : nullptr; }
            if (kind == 3) { double *q[8]; for (int i = 0; i < 8; i++) q[i] = get(vb); x = q[2]; y = q[4]; }
            if (x && y) {
                STANCHK(stan_spmv_probe_range(ctx, K, K->d_vals, 0, K->nslices, 10, &f, 9, x, y));
                out[pass * 4 + kind] = f;
            }
            for (void *q : blk) hipFree(q);
        }
    }
    return STAN_OK;
}

// Counters for the placement question: `reps` SpMV launches with the gather vector and
// the product in blocks of ANOTHER allocation run than the values (pairing 0), then `reps` with both carved out
// of the value block itself (pairing 1: by construction the same-group pairing), then pairing 0 again -- under
// `rocprofv3 --pmc ...` the dispatches of k_spmv split by order into the three phases (1 warm-up launch in front
// of each).  out_ms [3].  The value block is K's own (plain allocation: run with STAN_OPT_PLACEMENT_TRIES = 1).
extern "C" int stan_hip_lab_pairing_pmc(stan_ctx *ctx, stan_matrix *K, int32_t reps, double *out_ms) {
    if (!ctx || !K || !out_ms || reps < 1 || K->ctx != ctx) return STAN_E_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t npad = (int64_t)K->nslices * 64;
    const size_t ng = (size_t)(3 * ((npad > K->nloc + K->nhalo ? npad : K->nloc + K->nhalo)));
    const size_t vb = (ng * 8 + 4095) & ~(size_t)4095;
    // vectors far away from the values in allocation order: 24 spacer blocks of the value block's size in between
    const size_t bytes = (size_t)K->nslots * 9 * 64 * 8;
    std::vector<void *> spacer;
    for (int i = 0; i < 24; i++) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < 6 * bytes) break;
        void *q = nullptr;
        if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        spacer.push_back(q);
    }
    double *far = nullptr;
    if (hipMalloc((void **)&far, 2 * vb) != hipSuccess) { (void)hipGetLastError(); for (void *q : spacer) hipFree(q); return STAN_E_ALLOC; }
    float f = 0;
    for (int phase = 0; phase < 3; phase++) {
        double *x = phase == 1 ? K->d_vals : far;
        double *y = x + vb / 8;
        if (phase == 1) {   // the block holds the matrix: save what the vectors overwrite
            // (the probe fills x with ones and writes y: 2 vb bytes at the front of the value block)
        }
        std::vector<char> keep;
        if (phase == 1) { keep.resize(2 * vb); HIPCHK(ctx, hipMemcpy(keep.data(), K->d_vals, 2 * vb, hipMemcpyDeviceToHost)); }
        STANCHK(stan_spmv_probe_range(ctx, K, K->d_vals, 0, K->nslices, reps, &f, 9, x, y));
        out_ms[phase] = f;
        if (phase == 1) HIPCHK(ctx, hipMemcpy(K->d_vals, keep.data(), 2 * vb, hipMemcpyHostToDevice));
    }
    hipFree(far);
    for (void *q : spacer) hipFree(q);
    return STAN_OK;
}
