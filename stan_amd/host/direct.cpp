// direct.cpp -- CPU fallback for the reference's direct-solver options (SURVEY.md section 8f,
// rank 4): Analysis.LinSolver = "Cholesky" (the GUI's second choice, BOX_Analysis.xaml:23-24) and
// "LU".  They are NOT part of the GPU hot path; they exist so that a file the reference solves is
// not refused by the console driver.  Input is the reduced upper-triangle CRS the reference's
// alglib.sparsematrix holds (stan_hip_matrix_to_csr, upper_only = 1).
//
//   LinearSolver_Cholesky  SolverFunctions.cs:332-444
//     alglib.sparseconverttosks -> sparsecholeskyskyline(K, n, isupper=true) -> sparsecholeskysolvesks.
//     Restated from the ALGLIB manual text quoted in that file: in-place skyline factorisation
//     A = U^T U of the upper triangle, NO profile-reducing permutation (the rows keep the
//     reference's BFS order), False for a non-SPD matrix; the solve reports terminationtype > 0
//     for a solution and -3 with X filled by zeros otherwise.
//   LinearSolver_LU        SolverFunctions.cs:446-516
//     alglib.sparselu is handed the matrix ParallelAssembly_K built, which holds only col >= row
//     (SolverFunctions.cs:158): it factorises that UPPER-TRIANGULAR matrix as a general one, so
//     what the reference returns is the solution of triu(K) x = F, not of K x = F.  Unreachable
//     from the GUI (its combo box offers CG and Cholesky only).  Kept as the reference computes it:
//     a back substitution on the stored triangle (pivoting cannot change the solution of a
//     nonsingular triangular system).
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/stan_host.h"

extern "C" int stan_host_cholesky_skyline_solve(int64_t n, const int64_t *rowptr, const int32_t *col,
                                                const double *val, const double *b, double *x,
                                                int32_t *termination_type, int64_t *profile_entries) {
    if (n < 0 || !rowptr || (n > 0 && (!col || !val || !b || !x))) return STAN_HOST_E_ARG;
    if (termination_type) *termination_type = -3;
    // skyline of the upper triangle, column-wise: column j holds rows first[j] .. j
    std::vector<int64_t> first((size_t)n), cptr((size_t)n + 1, 0);
    for (int64_t j = 0; j < n; j++) first[(size_t)j] = j;
    for (int64_t i = 0; i < n; i++)
        for (int64_t q = rowptr[i]; q < rowptr[i + 1]; q++) {
            const int64_t j = col[q];
            if (j < i || j >= n) return STAN_HOST_E_ARG;   // upper triangle, columns in range
            if (i < first[(size_t)j]) first[(size_t)j] = i;
        }
    for (int64_t j = 0; j < n; j++) cptr[(size_t)j + 1] = cptr[(size_t)j] + (j - first[(size_t)j] + 1);
    if (profile_entries) *profile_entries = cptr[(size_t)n];
    if (cptr[(size_t)n] > ((int64_t)1 << 32)) return STAN_HOST_E_MEMORY;   // > 32 GiB of profile
    std::vector<double> sk;
    try { sk.assign((size_t)cptr[(size_t)n], 0.0); } catch (const std::bad_alloc &) { return STAN_HOST_E_MEMORY; }
    // column j is stored top (row first[j]) to bottom (the diagonal)
    auto at = [&](int64_t i, int64_t j) -> double & { return sk[(size_t)(cptr[(size_t)j] + (i - first[(size_t)j]))]; };
    for (int64_t i = 0; i < n; i++)
        for (int64_t q = rowptr[i]; q < rowptr[i + 1]; q++) at(i, col[q]) += val[q];
    for (int64_t i = 0; i < n; i++) x[i] = 0.0;   // "filled by zeros" on failure
    // A = U^T U, column by column: u_ij = (a_ij - sum_k u_ki u_kj) / u_ii, u_jj = sqrt(a_jj - sum_k u_kj^2)
    bool spd = true;
    for (int64_t j = 0; j < n && spd; j++) {
        const int64_t fj = first[(size_t)j];
        double *cj = &sk[(size_t)cptr[(size_t)j]];   // cj[i - fj] = entry (i, j)
        for (int64_t i = fj; i < j; i++) {
            const int64_t fi = first[(size_t)i];
            const int64_t k0 = fi > fj ? fi : fj;
            const double *ci = &sk[(size_t)cptr[(size_t)i]];
            double s = cj[i - fj];
            const double *pa = ci + (k0 - fi), *pb = cj + (k0 - fj);
            for (int64_t k = 0; k < i - k0; k++) s -= pa[k] * pb[k];
            cj[i - fj] = s / ci[i - fi];
        }
        double d = cj[j - fj];
        for (int64_t k = 0; k < j - fj; k++) d -= cj[k] * cj[k];
        if (!(d > 0.0) || !std::isfinite(d)) spd = false;
        else cj[j - fj] = std::sqrt(d);
    }
    if (!spd) return STAN_HOST_OK;   // reported through termination_type = -3, x = 0 (like the reference)
    // U^T y = b (forward), U x = y (backward)
    std::vector<double> y(b, b + n);
    for (int64_t j = 0; j < n; j++) {
        const int64_t fj = first[(size_t)j];
        const double *cj = &sk[(size_t)cptr[(size_t)j]];
        double s = y[(size_t)j];
        for (int64_t k = fj; k < j; k++) s -= cj[k - fj] * y[(size_t)k];
        y[(size_t)j] = s / cj[j - fj];
    }
    for (int64_t j = n - 1; j >= 0; j--) {
        const int64_t fj = first[(size_t)j];
        const double *cj = &sk[(size_t)cptr[(size_t)j]];
        const double xj = y[(size_t)j] / cj[j - fj];
        y[(size_t)j] = xj;
        for (int64_t k = fj; k < j; k++) y[(size_t)k] -= cj[k - fj] * xj;
    }
    for (int64_t i = 0; i < n; i++)
        if (!std::isfinite(y[(size_t)i])) return STAN_HOST_OK;
    memcpy(x, y.data(), (size_t)n * sizeof(double));
    if (termination_type) *termination_type = 1;
    return STAN_HOST_OK;
}

extern "C" int stan_host_lu_upper_solve(int64_t n, const int64_t *rowptr, const int32_t *col, const double *val,
                                        const double *b, double *x, int32_t *termination_type) {
    if (n < 0 || !rowptr || (n > 0 && (!col || !val || !b || !x))) return STAN_HOST_E_ARG;
    if (termination_type) *termination_type = -3;
    for (int64_t i = 0; i < n; i++) x[i] = 0.0;
    std::vector<double> y((size_t)n);
    for (int64_t i = n - 1; i >= 0; i--) {
        double s = b[i], d = 0.0;
        for (int64_t q = rowptr[i]; q < rowptr[i + 1]; q++) {
            const int64_t j = col[q];
            if (j < i || j >= n) return STAN_HOST_E_ARG;
            if (j == i) d += val[q];
            else s -= val[q] * y[(size_t)j];
        }
        if (d == 0.0 || !std::isfinite(d)) return STAN_HOST_OK;   // symbolically / numerically degenerate
        y[(size_t)i] = s / d;
    }
    memcpy(x, y.data(), (size_t)n * sizeof(double));
    if (termination_type) *termination_type = 1;
    return STAN_HOST_OK;
}
