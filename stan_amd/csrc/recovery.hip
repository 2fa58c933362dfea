// recovery.hip -- strain/stress recovery after the solve:
// Element.Recovery_Stress (Element.cs:211-246) + Update_StrainStress (:257-267) with the
// extrapolation table FE_Library.HEX8_ShapeFunctions (FE_Library.cs:105-116, 285-321).
//   eps_g = B_g u_e, sig_g = D eps_g at the 8 Gauss points, then per node
//   value_i = sum_g N[i][g] value_g,  N[i][g] = shape function g at xi = (+-1)/(1/sqrt 3).
// The reference reads the B_g it cached during K_Initial (~10 KB per element); here B_g is
// recomputed from the coordinates (embarrassingly parallel, 8 lanes per element = one per
// Gauss point, node extrapolation through wavefront shuffles).
// Compute_NodalForces (Element.cs:248-255) + the R assembly (Solver.cs:189-196) are the FORCES
// variant of the same kernel: f_e = sum_g B_g^T dS[g] det J_g w, with dS[g] the NODE-extrapolated
// stress of node g exactly as the reference indexes it; R[DOF] accumulates with fp64 atomics
// (the reference's own "+=" under Parallel.ForEach is an unsynchronised race).  The linear-static
// driver discards R (Solver.cs:199), so the console driver does not ask for it.
// HEX8_G1 makes the reference throw (N has one row, indexed by node: Element.cs:242 vs
// FE_Library.cs:77-81): reported as STAN_E_UNSUPPORTED with the element index.
#include "internal.h"
#include "hex8_device.h"

namespace {

// Round 5 (profiles/r05/k_recover_n148_kernel_trace_summary_r05_final.txt): the first form let each of the 8 lanes of an element load all 8 nodes' coordinates
// and displacements itself (56 gathers per lane) and store its 6 + 6 values 48 B apart: 1.73 ms at 148^3 for 3.2 GB =
// 0.23 of the HBM peak, bound by the address pipeline.  Now lane i of an element loads node i only (7 gathers) and the
// element's 48 values go round through LDS (one record per element, 49 doubles apart: no bank conflicts between the 8
// elements of a wave); the node extrapolation value_i = sum_g N[i][g] value_g uses the tensor structure of
// N[i][g] = prod_axis 1/2 (1 + s_i s_g sqrt 3) -- three butterfly stages (x: lane ^ 1, y: lane ^ 3, z: lane ^ 4 in CHEXA
// order) instead of an 8-term sum of shuffles; the results leave through LDS as full 512-B lines.
constexpr int REC = 50;   // doubles per element record in LDS (48 + 2: the 8 records of a wave start 36 banks apart -- no conflicts between them, and a record stays 16-B aligned: the reads pair up into ds_read_b128)

template <bool FORCES>
__global__ void __launch_bounds__(256)
k_recover(int64_t n_elem, const double *__restrict__ xyz, const double *__restrict__ disp, const int32_t *__restrict__ conn,
          const int32_t *__restrict__ elem_mat, const uint8_t *__restrict__ elem_type, const double *__restrict__ mat_lamG,
          double *__restrict__ strain, double *__restrict__ stress, long long *bad_elem, long long *g1_elem,
          const int32_t *__restrict__ node_dof, double *__restrict__ elem_forces, double *R) {
    __shared__ __attribute__((aligned(16))) double lds[4][8 * REC];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int el = lane >> 3, g = lane & 7;
    const int64_t e0 = ((int64_t)blockIdx.x * 4 + wv) * 8;   // first element of this wave
    const int64_t e = e0 + el;
    const bool valid = e < n_elem;
    double eps[6] = {0, 0, 0, 0, 0, 0}, sig[6] = {0, 0, 0, 0, 0, 0};
    double o[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, det = 0, px = 0, py = 0, pz = 0;
    bool live = false;  // a HEX8_G2 element of the batch
    int type = 0;
    int64_t nd = 0;
    double *rec = lds[wv] + el * REC;
    if (valid) {
        type = elem_type[e];
        nd = conn[e * 8 + g];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            rec[3 * g + c] = xyz[3 * nd + c];
            rec[24 + 3 * g + c] = disp[3 * nd + c];
        }
    }
    // wave-local exchange: the LDS executes one wave's instructions in order (as in k_spmv_fold)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (valid) {
        if (type != STAN_HEX8_G2) {
            if (g == 0) atomicMin(g1_elem, (long long)e);
        } else {
            live = true;
            const double *u = rec + 24;   // the element's record stays in LDS: coordinates [0, 24), displacements [24, 48)
            det = hex8_gp_setup(rec, type, g, o);
            if (det == 0.0) atomicMin(bad_elem, (long long)e);
            const double gl = hex8_gauss_loc(type);
            px = hex8_sign(HEX8_SX, g) * gl; py = hex8_sign(HEX8_SY, g) * gl;
            pz = hex8_sign(HEX8_SZ, g) * gl;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                double gr[3];
                hex8_grad(o, i, px, py, pz, gr);
                // BL0 rows (Element.cs:316-324): xx, yy, zz, xy, yz, xz
                eps[0] += gr[0] * u[3 * i];
                eps[1] += gr[1] * u[3 * i + 1];
                eps[2] += gr[2] * u[3 * i + 2];
                eps[3] += gr[1] * u[3 * i] + gr[0] * u[3 * i + 1];
                eps[4] += gr[2] * u[3 * i + 1] + gr[1] * u[3 * i + 2];
                eps[5] += gr[2] * u[3 * i] + gr[0] * u[3 * i + 2];
            }
            const int32_t m = elem_mat[e];
            const double lam = mat_lamG[2 * m], G = mat_lamG[2 * m + 1];
            const double tr = lam * (eps[0] + eps[1] + eps[2]);
            sig[0] = tr + 2 * G * eps[0];
            sig[1] = tr + 2 * G * eps[1];
            sig[2] = tr + 2 * G * eps[2];
            sig[3] = G * eps[3]; sig[4] = G * eps[4]; sig[5] = G * eps[5];
        }
    }
    // node i = this lane's index within the element; N[i][k] = prod over the axes of 1/2 (1 + s_i s_k sqrt 3): a when node
    // and Gauss point lie on the same side of the axis, b otherwise (FE_Library.cs:105-116, 285-321 evaluated at
    // xi = +-sqrt 3); the partner across an axis is lane ^ 1 (xi), ^ 3 (eta), ^ 4 (zeta) in the CHEXA order of HEX8_S*
    const int i = g;
    const double ca = 0.5 * (1.0 + 1.7320508075688772935), cb = 0.5 * (1.0 - 1.7320508075688772935);
    double ne[6], ns[6];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        double a = eps[c], b = sig[c];
        a = ca * a + cb * __shfl_xor(a, 1, 64); b = ca * b + cb * __shfl_xor(b, 1, 64);
        a = ca * a + cb * __shfl_xor(a, 3, 64); b = ca * b + cb * __shfl_xor(b, 3, 64);
        a = ca * a + cb * __shfl_xor(a, 4, 64); b = ca * b + cb * __shfl_xor(b, 4, 64);
        ne[c] = a; ns[c] = b;
    }
    if (strain) {
        // the wave's 8 x 48 values of each array are contiguous in memory: through LDS, out as six 512-B lines
        double *stg = lds[wv];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int c = 0; c < 6; c++) stg[lane * 6 + c] = q ? ns[c] : ne[c];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double *dst = (q ? stress : strain) + e0 * 48;
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const int idx = j * 64 + lane;
                if (e0 + idx / 48 < n_elem) __builtin_nontemporal_store(stg[idx], dst + idx);
            }
        }
    }
    if (FORCES) {
        // lane g: B_g^T ns * det J_g * w (w = 1 for HEX8_G2), then the sum over the 8 lanes of
        // the element; node a's three components end up on lane a
        const double sc = live ? det : 0.0;
        double mine[3] = {0, 0, 0};
#pragma unroll
        for (int a = 0; a < 8; a++) {
            double gr[3] = {0, 0, 0};
            if (live) hex8_grad(o, a, px, py, pz, gr);
            double f[3];
            f[0] = (gr[0] * ns[0] + gr[1] * ns[3] + gr[2] * ns[5]) * sc;
            f[1] = (gr[1] * ns[1] + gr[0] * ns[3] + gr[2] * ns[4]) * sc;
            f[2] = (gr[2] * ns[2] + gr[1] * ns[4] + gr[0] * ns[5]) * sc;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                double v = f[c];
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                v += __shfl_xor(v, 4, 64);
                if (a == i) mine[c] = v;
            }
        }
        if (live) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                if (elem_forces) elem_forces[e * 24 + 3 * i + c] = mine[c];
                if (R) unsafeAtomicAdd(&R[node_dof[3 * nd + c]], mine[c]);
            }
        }
    }
}

}  // namespace

int stan_recover_device(stan_ctx *ctx, int64_t n_nodes, const double *d_xyz, const double *d_disp,
                        int64_t n_elem, const int32_t *d_conn, const int32_t *d_elem_mat,
                        const uint8_t *d_elem_type, int32_t n_mat, const double *mat_E_nu,
                        double *d_strain, double *d_stress, const int32_t *d_node_dof,
                        double *d_elem_forces, double *d_R) {
    (void)n_nodes;
    const bool forces = d_elem_forces || d_R;
    if (n_elem <= 0) return STAN_OK;
    std::vector<double> lamG(2 * (size_t)n_mat);
    for (int m = 0; m < n_mat; m++) stan_lame(mat_E_nu[2 * m], mat_E_nu[2 * m + 1], &lamG[2 * m], &lamG[2 * m + 1]);
    double *d_lamG;
    STANCHK(stan_dmalloc(ctx, &d_lamG, lamG.size()));
    hipStream_t st = ctx->stream;
    long long init[2] = {0x7fffffffffffffffLL, 0x7fffffffffffffffLL};
    hipError_t e1 = hipMemcpyAsync(d_lamG, lamG.data(), lamG.size() * 8, hipMemcpyHostToDevice, st);
    hipError_t e2 = hipMemcpyAsync(ctx->d_status + SS_BAD_ELEM, init, 16, hipMemcpyHostToDevice, st);
    const dim3 grid((unsigned)((n_elem + 31) / 32)), block(256);   // 8 lanes per element, 8 elements per wave
    if (forces)
        hipLaunchKernelGGL(k_recover<true>, grid, block, 0, st, n_elem, d_xyz, d_disp, d_conn, d_elem_mat,
                           d_elem_type, d_lamG, d_strain, d_stress, (long long *)(ctx->d_status + SS_BAD_ELEM),
                           (long long *)(ctx->d_status + SS_AUX), d_node_dof, d_elem_forces, d_R);
    else
        hipLaunchKernelGGL(k_recover<false>, grid, block, 0, st, n_elem, d_xyz, d_disp, d_conn, d_elem_mat,
                           d_elem_type, d_lamG, d_strain, d_stress, (long long *)(ctx->d_status + SS_BAD_ELEM),
                           (long long *)(ctx->d_status + SS_AUX), nullptr, nullptr, nullptr);
    hipError_t e3 = hipGetLastError();
    hipError_t e4 = hipMemcpyAsync(ctx->h_status + SS_BAD_ELEM, ctx->d_status + SS_BAD_ELEM, 16, hipMemcpyDeviceToHost, st);
    hipError_t e5 = hipStreamSynchronize(st);
    stan_dfree(ctx, d_lamG);
    for (hipError_t e : {e1, e2, e3, e4, e5})
        if (e != hipSuccess) { ctx->err = std::string("recover: ") + hipGetErrorString(e); return STAN_E_HIP; }
    if (ctx->h_status[SS_AUX] != init[0]) {
        ctx->bad_elem = ctx->h_status[SS_AUX];
        ctx->err = "stress recovery: element " + std::to_string(ctx->bad_elem) +
                   " is HEX8_G1 (the reference throws: N has one row, Element.cs:242)";
        return STAN_E_UNSUPPORTED;
    }
    if (ctx->h_status[SS_BAD_ELEM] != init[0]) {
        ctx->bad_elem = ctx->h_status[SS_BAD_ELEM];
        ctx->err = "det J == 0 in element " + std::to_string(ctx->bad_elem);
        return STAN_E_DETJ;
    }
    return STAN_OK;
}
