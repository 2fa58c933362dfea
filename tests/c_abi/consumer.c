/* consumer.c -- a plain C99 host of the two C-ABI libraries, written against include/ only:
 * the same call sequence INTEGRATION.md's P/Invoke shim performs (Solver.cs:46, 104-178):
 *   AssignDOF -> nDOF_reduction -> F -> ParallelAssembly_K -> LinearSolver_CG -> write-back.
 * Mesh: an n x n x n cube of unit HEX8_G2 elements, clamp x = 0, PointLoad (0,0,50) on x = n.
 * usage: consumer <n> [out.bin]   prints sizes and the CG report; writes disp (fp64) if asked.
 * Without a GPU stan_hip_init fails loudly (non-zero, message) and the program exits 3. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "stan_hip.h"
#include "stan_host.h"

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 2;
    const int m = n + 1;
    const int64_t n_nodes = (int64_t)m * m * m, n_elem = (int64_t)n * n * n, n_dof = 3 * n_nodes;
    double *xyz = malloc(sizeof(double) * 3 * (size_t)n_nodes);
    int32_t *conn = malloc(sizeof(int32_t) * 8 * (size_t)n_elem);
    for (int k = 0; k < m; k++)
        for (int j = 0; j < m; j++)
            for (int i = 0; i < m; i++) {
                const int64_t id = i + (int64_t)m * (j + (int64_t)m * k);
                xyz[3 * id] = i; xyz[3 * id + 1] = j; xyz[3 * id + 2] = k;
            }
#define NID(a, b, c) ((int32_t)((a) + m * ((b) + m * (c))))
    for (int k = 0; k < n; k++)
        for (int j = 0; j < n; j++)
            for (int i = 0; i < n; i++) {
                int32_t *c = conn + 8 * (i + (int64_t)n * (j + (int64_t)n * k));
                c[0] = NID(i, j, k); c[1] = NID(i + 1, j, k); c[2] = NID(i + 1, j + 1, k); c[3] = NID(i, j + 1, k);
                c[4] = NID(i, j, k + 1); c[5] = NID(i + 1, j, k + 1); c[6] = NID(i + 1, j + 1, k + 1);
                c[7] = NID(i, j + 1, k + 1);
            }
    /* Database.AssignDOF */
    int32_t *node_index = malloc(sizeof(int32_t) * (size_t)n_nodes);
    int32_t *node_dof = malloc(sizeof(int32_t) * 3 * (size_t)n_nodes);
    int rc = stan_host_assign_dof(n_nodes, n_elem, conn, node_index, node_dof);
    if (rc) { fprintf(stderr, "assign_dof: %d\n", rc); return 1; }
    /* boundary conditions */
    const int64_t n_face = (int64_t)m * m;
    int32_t *spc = malloc(sizeof(int32_t) * (size_t)n_face), *ld = malloc(sizeof(int32_t) * (size_t)n_face);
    double *spc_v = malloc(sizeof(double) * 3 * (size_t)n_face), *ld_v = malloc(sizeof(double) * 3 * (size_t)n_face);
    int64_t q = 0;
    for (int k = 0; k < m; k++)
        for (int j = 0; j < m; j++, q++) {
            spc[q] = NID(0, j, k); ld[q] = NID(n, j, k);
            spc_v[3 * q] = spc_v[3 * q + 1] = spc_v[3 * q + 2] = 1.0;
            ld_v[3 * q] = 0.0; ld_v[3 * q + 1] = 0.0; ld_v[3 * q + 2] = 50.0;
        }
    int32_t *red = malloc(sizeof(int32_t) * (size_t)n_dof);
    int64_t n_fixed = 0;
    rc = stan_host_dof_reduction(n_dof, node_dof, n_face, spc, spc_v, red, &n_fixed);
    if (rc) { fprintf(stderr, "dof_reduction: %d\n", rc); return 1; }
    const int64_t N = n_dof - n_fixed;
    double *F = calloc((size_t)N, sizeof(double)), *U = calloc((size_t)N, sizeof(double));
    rc = stan_host_load_vector(n_dof, node_dof, red, n_face, ld, ld_v, F);
    if (rc) { fprintf(stderr, "load_vector: %d\n", rc); return 1; }
    printf("nodes %lld elements %lld nDOF %lld fixed %lld N %lld\n", (long long)n_nodes, (long long)n_elem,
           (long long)n_dof, (long long)n_fixed, (long long)N);

    /* the hot path */
    stan_ctx *ctx = NULL;
    rc = stan_hip_init(0, &ctx);
    if (rc) {
        fprintf(stderr, "stan_hip_init failed (%d): %s\n", rc, stan_hip_last_error(NULL));
        return 3;
    }
    int32_t *elem_mat = calloc((size_t)n_elem, sizeof(int32_t));
    uint8_t *elem_type = malloc((size_t)n_elem);
    memset(elem_type, STAN_HEX8_G2, (size_t)n_elem);
    const double mat[2] = {210000.0, 0.3};
    stan_matrix *K = NULL;
    rc = stan_hip_assemble_hex8(ctx, n_nodes, xyz, node_dof, n_elem, conn, elem_mat, elem_type, 1, mat, n_dof,
                                red, &K);
    if (rc) { fprintf(stderr, "assemble: %d %s\n", rc, stan_hip_last_error(ctx)); return 1; }
    int32_t type = 0, its = 0;
    double rel = 0;
    rc = stan_hip_cg_solve(ctx, K, F, 1e-10, 0, STAN_PREC_FP64, U, &type, &its, &rel);
    if (rc) { fprintf(stderr, "cg_solve: %d %s\n", rc, stan_hip_last_error(ctx)); return 1; }
    printf("CG %s (type %d) iterations %d rel_residual %.3e\n", type == 1 || type == 7 ? "NORMAL" : "ERROR", type,
           its, rel);
    stan_hip_matrix_free(K);
    double *disp = malloc(sizeof(double) * 3 * (size_t)n_nodes);
    rc = stan_host_nodal_displacements(n_nodes, node_dof, red, U, disp);
    if (rc) { fprintf(stderr, "nodal_displacements: %d\n", rc); return 1; }
    double *strain = malloc(sizeof(double) * 48 * (size_t)n_elem), *stress = malloc(sizeof(double) * 48 * (size_t)n_elem);
    rc = stan_hip_recover_hex8(ctx, n_nodes, xyz, disp, n_elem, conn, elem_mat, elem_type, 1, mat, strain, stress);
    if (rc) { fprintf(stderr, "recover: %d %s\n", rc, stan_hip_last_error(ctx)); return 1; }
    stan_hip_destroy(ctx);
    double tip = 0;
    for (int64_t i = 0; i < n_nodes; i++)
        if (disp[3 * i + 2] > tip) tip = disp[3 * i + 2];
    printf("max uz %.12e  stress_xx[0][0] %.12e\n", tip, stress[0]);
    if (argc > 2) {
        FILE *f = fopen(argv[2], "wb");
        if (!f || fwrite(disp, sizeof(double), 3 * (size_t)n_nodes, f) != 3 * (size_t)n_nodes) return 1;
        fclose(f);
    }
    return 0;
}
