/*
 * stan_oracle.h -- CPU restatement of STAN's linear-static hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (galuszkm/STAN, C#/.NET 4.6.1) ships no
 * tests, golden vectors or fixtures, cannot be built in this image (no .NET
 * toolchain) and its CG lives in alglib.net 3.16.0, which is not vendored.
 * The functions below follow the C# sources statement by statement (each
 * cites the file:line it restates); the CG restates ALGLIB's published
 * lincg algorithm.  They are pinned only against independent analytic and
 * SciPy results (tests/test_oracle_*.py), never against reference output.
 *
 * Build with -ffp-contract=off so that a*b+c is two roundings as in the
 * .NET JIT (no FMA contraction).
 */
#ifndef STAN_ORACLE_H
#define STAN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element type codes used across the repo (Element.Type strings) */
#define STAN_HEX8_G1 1 /* "HEX8_G1": 1 Gauss point, weight 8  (FE_Library.cs:63-89)  */
#define STAN_HEX8_G2 2 /* "HEX8_G2": 8 Gauss points, weight 1 (FE_Library.cs:91-131) */

/* FE_Library.cs:206-276  HEX8_Diff_ShapeFunctions at Gauss point g of `type`.
 * out = 3x8 row-major.  Returns number of Gauss points of the type, <0 on bad type. */
int stan_oracle_dn_dlocal(int type, int g, double out[24]);
/* FE_Library.cs:285-321 extrapolation table N[i][g] (8x8 for G2) */
int stan_oracle_extrap_N(int type, double out[64]);
/* Material.cs:31-56 SetElastic -> 6x6 row-major D */
void stan_oracle_material_D(double E, double nu, double D[36]);

/* Element.cs:118-155 K_Initial (order-faithful MatrixST arithmetic).
 * xyz8: 8x3 row-major nodal coordinates in NList order; K: 24x24 row-major.
 * Returns 0, or -1 if det J == 0 at a Gauss point (MatrixST.cs:315-318 throws). */
int stan_oracle_ke_hex8(const double xyz8[24], const double D[36], int type, double K[576]);

/* Database.cs:140-234 AssignDOF, literal restatement (neighbour lists, first
 * node rule, FIFO with duplicates).  conn: n_elem x 8 node *indices* (position
 * in NodeLib order).  node_index_out[i] = BFS index of node i (DOF = 3*idx+{0,1,2},
 * Node.cs:218-223).  Returns 0; -2 no start node (Database.cs:179-196 leaves
 * FirstNode=0 -> KeyNotFound); -3 queue exhausted (disconnected mesh,
 * Database.cs:218 ArgumentOutOfRange). */
int stan_oracle_assign_dof(int64_t n_nodes, int64_t n_elem, const int32_t *conn,
                           int32_t *node_index_out);

/* Solver.cs:121-132: red[i] = -1 if fixed else #fixed below i. fixed = 0/1 flags per DOF.
 * Returns number of fixed DOFs. */
int64_t stan_oracle_dof_reduction(int64_t n_dof, const uint8_t *fixed, int32_t *red);

/* ---- sparse matrix: ALGLIB-style hash accumulate -> CRS (upper triangle) ---- */
typedef struct stan_oracle_crs {
    int64_t n;
    int64_t nnz;
    int64_t *ridx; /* n+1 */
    int32_t *idx;  /* nnz, ascending within a row */
    double *vals;  /* nnz */
} stan_oracle_crs;

/* SolverFunctions.cs:117-180 ParallelAssembly_K (sequential element order) +
 * sparsecreate/sparseadd/sparseconverttocrs.  node_dof: n_nodes x 3 (Node.DOF).
 * elem_mat: index into mat_E_nu (pairs E,nu); elem_type: STAN_HEX8_G1/G2.
 * n_threads>1 computes K_e in parallel (like TPL) but scatters serially, in
 * element order.  Returns 0 or a negative error (-1: det J==0, *bad_elem set). */
int stan_oracle_assemble(int64_t n_nodes, const double *xyz, const int32_t *node_dof,
                         int64_t n_elem, const int32_t *conn, const int32_t *elem_mat,
                         const uint8_t *elem_type, int32_t n_mat, const double *mat_E_nu,
                         int64_t n_dof, const int32_t *red, int n_threads,
                         stan_oracle_crs *out, int64_t *bad_elem);
void stan_oracle_crs_free(stan_oracle_crs *m);

/* ALGLIB sparsesmv(isupper=true): y = A x from the upper triangle. */
void stan_oracle_smv_upper(const stan_oracle_crs *A, const double *x, double *y);

/* SolverFunctions.cs:270-330 LinearSolver_CG == alglib lincgcreate/lincgsetcond/
 * lincgsolvesparse/lincgresults (diagonal scaling, r refresh every 10 its,
 * merit-function stop).  Returns 0.  term/iters/nmv as in lincgreport. */
int stan_oracle_cg(const stan_oracle_crs *A, const double *b, double epsf, int32_t maxits,
                   double *x, int32_t *terminationtype, int32_t *iterations, int32_t *nmv,
                   double *rel_residual_scaled);

/* Same with the two switches libstan_hip.so exposes as STAN_OPT_CG_MERIT_STOP / _RUPDATE
 * (merit_stop=1, itsbeforerupdate=10 is ALGLIB's behaviour). */
int stan_oracle_cg_opt(const stan_oracle_crs *A, const double *b, double epsf, int32_t maxits,
                       int merit_stop, int itsbeforerupdate, double *x, int32_t *terminationtype,
                       int32_t *iterations, int32_t *nmv, double *rel_residual_scaled);

/* n > 1: the CG's matrix-vector product runs on n OpenMP threads over an expanded full CRS
 * (a labelled, fairer CPU number; alglib's own product is serial).  Default 1. */
void stan_oracle_set_mv_threads(int n);

/* SolverFunctions.cs:520-538 Include_BC_DOF */
void stan_oracle_include_bc(int64_t n_dof, const int32_t *red, const double *U, double *U_full);

/* Element.cs:211-246 Recovery_Stress + :257-267 Update_StrainStress for one element:
 * dU 24 (node-major), out strain/stress 8x6 row-major (node x {xx,yy,zz,xy,yz,xz}).
 * Returns 0; -1 det J==0; -4 for HEX8_G1 (reference indexes N[i][g] with N.Count==1
 * and throws, Element.cs:242 vs FE_Library.cs:77-81). */
int stan_oracle_recover_hex8(const double xyz8[24], const double D[36], int type,
                             const double dU[24], double strain[48], double stress[48]);

/* Element.cs:248-255 Compute_NodalForces for one element, from the node-extrapolated stress
 * rows stan_oracle_recover_hex8 returned (the reference indexes that node list by Gauss point
 * number -- kept).  forces: 24, node-major.  The driver sums them into R[DOF] (Solver.cs:189-196)
 * and discards R in the linear-static path (Solver.cs:199). */
int stan_oracle_nodal_forces_hex8(const double xyz8[24], int type, const double stress_nodes[48],
                                  double forces[24]);

#ifdef __cplusplus
}
#endif
#endif
