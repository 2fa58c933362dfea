// hex8_device.h -- HEX8 element math shared by the assembly and recovery kernels.
//
// Restates (in closed form, fp64) what Element.K_Initial computes (Element.cs:118-155):
//   K_e = sum_g (B^T D B)(det J_g * w),  J_g = dN_dLocal[g] * X (Element.cs:274-292),
//   dN = J^-1 dN_dLocal (Element.cs:130), B = BL0 (Element.cs:297-328; BL1 == 0 in
//   linear statics), D isotropic from (E, nu) (Material.cs:31-56).
// For isotropic D the 3x3 block that couples local nodes a and b is
//   K_ab[m][n] = sum_g c_g ( lambda * ga[m]*gb[n] + G * ga[n]*gb[m] + delta_mn G (ga . gb) )
// with ga = grad N_a, gb = grad N_b at Gauss point g and c_g = det J_g * w.
// Same algebra as the reference, different rounding order (parity bar: <= 1e-13 relative
// to max|K_e|).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// Natural-coordinate sign tables of FE_Library.cs:225-235 / :121-128 packed as bit masks:
// bit i set <=> coordinate of node (or Gauss point) i is +1.
#define HEX8_SX 0x66u  // xi  : - + + - - + + -
#define HEX8_SY 0xCCu  // eta : - - + + - - + +
#define HEX8_SZ 0xF0u  // zeta: - - - - + + + +

__device__ __forceinline__ double hex8_sign(unsigned mask, int i) {
    return ((mask >> i) & 1u) ? 1.0 : -1.0;
}

// dN_i/d(xi,eta,zeta) at natural point (px,py,pz): FE_Library.cs:246-273 factorised,
// e.g. dN1/dxi = 1/8(-1+eta+zeta-eta*zeta) = 1/8 * xi_1 * (1+eta_1*eta)(1+zeta_1*zeta).
__device__ __forceinline__ void hex8_dnl(int i, double px, double py, double pz, double d[3]) {
    const double sx = hex8_sign(HEX8_SX, i), sy = hex8_sign(HEX8_SY, i),
                 sz = hex8_sign(HEX8_SZ, i);
    const double fx = 1.0 + sx * px, fy = 1.0 + sy * py, fz = 1.0 + sz * pz;
    d[0] = 0.125 * sx * fy * fz;
    d[1] = 0.125 * sy * fx * fz;
    d[2] = 0.125 * sz * fx * fy;
}

// Gauss point location (FE_Library.cs:75,103) and weight (:72,:100) of point g for `type`.
// HEX8_G1 has one point at the origin with weight 8: points g>0 get weight 0.
__device__ __forceinline__ double hex8_gauss_loc(int type) {
    return type == STAN_HEX8_G2 ? 0.57735026918962576451 /* sqrt(1/3) */ : 0.0;
}
__device__ __forceinline__ double hex8_gauss_weight(int type, int g) {
    return type == STAN_HEX8_G2 ? 1.0 : (g == 0 ? 8.0 : 0.0);
}

// Jacobian at Gauss point g, its inverse (adjugate/det, MatrixST.cs:294-319) and
// c = det J * w.  x: 8 nodes x 3 coordinates (any addressable memory).
// out[0..8] = J^-1 row-major, out[9] = c.  Returns det J.
__device__ __forceinline__ double hex8_gp_setup(const double *x, int type, int g, double out[10]) {
    const double gl = hex8_gauss_loc(type);
    const double px = hex8_sign(HEX8_SX, g) * gl, py = hex8_sign(HEX8_SY, g) * gl,
                 pz = hex8_sign(HEX8_SZ, g) * gl;
    double J[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        double d[3];
        hex8_dnl(i, px, py, pz, d);
        const double x0 = x[3 * i], x1 = x[3 * i + 1], x2 = x[3 * i + 2];
        J[0] += d[0] * x0; J[1] += d[0] * x1; J[2] += d[0] * x2;
        J[3] += d[1] * x0; J[4] += d[1] * x1; J[5] += d[1] * x2;
        J[6] += d[2] * x0; J[7] += d[2] * x1; J[8] += d[2] * x2;
    }
    // MatrixST.cs:270-287 Det3
    const double det = J[0] * J[4] * J[8] + J[3] * J[7] * J[2] + J[6] * J[1] * J[5] -
                       J[2] * J[4] * J[6] - J[0] * J[5] * J[7] - J[8] * J[1] * J[3];
    const double X = 1.0 / det;
    out[0] = X * (J[4] * J[8] - J[5] * J[7]);
    out[1] = X * (J[2] * J[7] - J[1] * J[8]);
    out[2] = X * (J[1] * J[5] - J[2] * J[4]);
    out[3] = X * (J[5] * J[6] - J[3] * J[8]);
    out[4] = X * (J[0] * J[8] - J[2] * J[6]);
    out[5] = X * (J[2] * J[3] - J[0] * J[5]);
    out[6] = X * (J[3] * J[7] - J[4] * J[6]);
    out[7] = X * (J[1] * J[6] - J[0] * J[7]);
    out[8] = X * (J[0] * J[4] - J[1] * J[3]);
    out[9] = det * hex8_gauss_weight(type, g);
    return det;
}

// Global gradient of shape function i at Gauss point g: grad = J^-1 * dNl[:, i].
__device__ __forceinline__ void hex8_grad(const double *inv, int i, double px, double py,
                                          double pz, double gr[3]) {
    double d[3];
    hex8_dnl(i, px, py, pz, d);
    gr[0] = inv[0] * d[0] + inv[1] * d[1] + inv[2] * d[2];
    gr[1] = inv[3] * d[0] + inv[4] * d[1] + inv[5] * d[2];
    gr[2] = inv[6] * d[0] + inv[7] * d[1] + inv[8] * d[2];
}

// The (a,b) 3x3 block of K_e over the element's Gauss points, in the M-form (round 4):
//   M_ab = sum_g (c_g grad N_a)(grad N_b)^T          9 fused multiply-adds per Gauss point
//   K_ab = lambda M + G M^T + G tr(M) I               once, after the loop
// -- the same algebra as before (K_ab[m][n] = sum_g c_g (lambda ga[m] gb[n] + G ga[n] gb[m] + delta_mn G ga.gb)) with a
// third of the instructions per Gauss point.  Every product and sum is spelled out with fma() so that the kernels that
// share these helpers (row-owner gather, its wide-row form, the colour scatter, k_ke_batch) round alike whatever the
// compiler would contract on its own.
// grad = J^-1 d, d = dN/d(xi,eta,zeta) of one node
__device__ __forceinline__ void hex8_inv_times(const double *inv, const double d[3], double gr[3]) {
    gr[0] = fma(inv[2], d[2], fma(inv[1], d[1], inv[0] * d[0]));
    gr[1] = fma(inv[5], d[2], fma(inv[4], d[1], inv[3] * d[0]));
    gr[2] = fma(inv[8], d[2], fma(inv[7], d[1], inv[6] * d[0]));
}
// w = c * grad N_a at the Gauss point whose {J^-1, c} is q[0..9]
__device__ __forceinline__ void hex8_wgrad(const double *q, int a, double px, double py, double pz, double w[3]) {
    double d[3], ga[3];
    hex8_dnl(a, px, py, pz, d);
    hex8_inv_times(q, d, ga);
    w[0] = q[9] * ga[0]; w[1] = q[9] * ga[1]; w[2] = q[9] * ga[2];
}
// M += w (J^-1 d_b)^T
__device__ __forceinline__ void hex8_m_accum(const double *inv, const double w[3], const double d[3], double M[9]) {
    double gb[3];
    hex8_inv_times(inv, d, gb);
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int n = 0; n < 3; n++) M[3 * m + n] = fma(w[m], gb[n], M[3 * m + n]);
}
__device__ __forceinline__ void hex8_k_from_m(const double M[9], double lam, double G, double k[9]) {
    const double gtr = G * ((M[0] + M[4]) + M[8]);
#pragma unroll
    for (int m = 0; m < 3; m++)
#pragma unroll
        for (int n = 0; n < 3; n++) {
            const double v = fma(lam, M[3 * m + n], G * M[3 * n + m]);
            k[3 * m + n] = m == n ? v + gtr : v;
        }
}
// gp: per Gauss point 10 doubles {J^-1, c} with stride `gstride` doubles between points.
#ifndef STAN_GP_UNROLL
#define STAN_GP_UNROLL 2
#endif
__device__ __forceinline__ void hex8_block_ab(const double *gp, int gstride, int type, int a,
                                              int b, double lam, double G, double k[9]) {
    const double gl = hex8_gauss_loc(type);
    double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll STAN_GP_UNROLL
    for (int g = 0; g < 8; g++) {
        const double *q = gp + g * gstride;
        const double px = hex8_sign(HEX8_SX, g) * gl, py = hex8_sign(HEX8_SY, g) * gl,
                     pz = hex8_sign(HEX8_SZ, g) * gl;
        double w[3], d[3];
        hex8_wgrad(q, a, px, py, pz, w);
        hex8_dnl(b, px, py, pz, d);
        hex8_m_accum(q, w, d, M);
    }
    hex8_k_from_m(M, lam, G, k);
}

// Lame constants exactly as Material.SetElastic forms them (Material.cs:39-40).
__host__ __device__ __forceinline__ void stan_lame(double E, double nu, double *lam, double *G) {
    *lam = (E * nu) / ((1 - 2 * nu) * (1 + nu));
    *G = (0.5 * E) / (1 + nu);
}
