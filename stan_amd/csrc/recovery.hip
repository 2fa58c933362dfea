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

template <bool FORCES>
__global__ void __launch_bounds__(256)
k_recover(int64_t n_elem, const double *xyz, const double *disp, const int32_t *conn,
          const int32_t *elem_mat, const uint8_t *elem_type, const double *mat_lamG,
          double *strain, double *stress, long long *bad_elem, long long *g1_elem,
          const int32_t *node_dof, double *elem_forces, double *R) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t e = t >> 3;
    const int g = (int)(t & 7);
    const int lane = threadIdx.x & 63;
    const bool valid = e < n_elem;
    double eps[6] = {0, 0, 0, 0, 0, 0}, sig[6] = {0, 0, 0, 0, 0, 0};
    double o[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, det = 0, px = 0, py = 0, pz = 0;
    bool live = false;  // a HEX8_G2 element of the batch
    if (valid) {
        const int type = elem_type[e];
        if (type != STAN_HEX8_G2) {
            if (g == 0) atomicMin(g1_elem, (long long)e);
        } else {
            live = true;
            double x[24], u[24];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int64_t nd = conn[e * 8 + i];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    x[3 * i + c] = xyz[3 * nd + c];
                    u[3 * i + c] = disp[3 * nd + c];
                }
            }
            det = hex8_gp_setup(x, type, g, o);
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
    // node i = this lane's index within the element; N[i][k] = 1/8 prod (1 + s_i s_k sqrt 3)
    const int i = g;
    const double r3 = 1.7320508075688772935;  // 1 / GaussLocation
    double ne[6] = {0, 0, 0, 0, 0, 0}, ns[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const double N = 0.125 * (1 + hex8_sign(HEX8_SX, i) * hex8_sign(HEX8_SX, k) * r3) *
                         (1 + hex8_sign(HEX8_SY, i) * hex8_sign(HEX8_SY, k) * r3) *
                         (1 + hex8_sign(HEX8_SZ, i) * hex8_sign(HEX8_SZ, k) * r3);
        const int src = (lane & ~7) | k;
#pragma unroll
        for (int c = 0; c < 6; c++) {
            ne[c] += N * __shfl(eps[c], src, 64);
            ns[c] += N * __shfl(sig[c], src, 64);
        }
    }
    if (valid && strain) {
#pragma unroll
        for (int c = 0; c < 6; c++) {
            strain[e * 48 + i * 6 + c] = ne[c];
            stress[e * 48 + i * 6 + c] = ns[c];
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
            const int64_t nd = conn[e * 8 + i];
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
    const int64_t nthreads = n_elem * 8;
    const dim3 grid((unsigned)((nthreads + 255) / 256)), block(256);
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
