/*
 * stan_oracle.c -- CPU restatement of STAN's linear-static hot path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see stan_oracle.h).
 *
 * Every function cites the reference file:line it restates.  The tiny
 * dense algebra keeps MatrixST's exact operation order (zero-initialised
 * accumulators, ascending inner index, allocate-per-op value semantics).
 */
#include "stan_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* MatrixST (MatrixST.cs:15-26): row-major double[] with Rows, Cols.          */
typedef struct {
    int rows, cols;
    double m[24 * 24];
} mst;

static void mst_zero(mst *a, int r, int c) {
    a->rows = r;
    a->cols = c;
    memset(a->m, 0, sizeof(double) * (size_t)(r * c));
}
#define G(a, i, j) ((a)->m[(i) * (a)->cols + (j)])

/* MatrixST.cs:404-427 operator*: C zero-init, C[i,j] += A[i,k]*B[k,j], k ascending */
static void mst_mul(const mst *A, const mst *B, mst *C) {
    mst_zero(C, A->rows, B->cols);
    for (int i = 0; i < A->rows; i++)
        for (int j = 0; j < B->cols; j++)
            for (int k = 0; k < A->cols; k++) G(C, i, j) += G(A, i, k) * G(B, k, j);
}
/* MatrixST.cs:435-454 operator+: C zero-init, C[i,j] += A[i,j] + B[i,j] */
static void mst_add(const mst *A, const mst *B, mst *C) {
    mst_zero(C, A->rows, A->cols);
    for (int i = 0; i < A->rows; i++)
        for (int j = 0; j < A->cols; j++) G(C, i, j) += G(A, i, j) + G(B, i, j);
}
/* MatrixST.cs:251-263 Transpose */
static void mst_transpose(const mst *A, mst *C) {
    mst_zero(C, A->cols, A->rows);
    for (int r = 0; r < A->rows; r++)
        for (int c = 0; c < A->cols; c++) G(C, c, r) = G(A, r, c);
}
/* MatrixST.cs:270-287 Det3 */
static double mst_det3(const mst *A) {
    return G(A, 0, 0) * G(A, 1, 1) * G(A, 2, 2) + G(A, 1, 0) * G(A, 2, 1) * G(A, 0, 2) +
           G(A, 2, 0) * G(A, 0, 1) * G(A, 1, 2) - G(A, 0, 2) * G(A, 1, 1) * G(A, 2, 0) -
           G(A, 0, 0) * G(A, 1, 2) * G(A, 2, 1) - G(A, 2, 2) * G(A, 0, 1) * G(A, 1, 0);
}
/* MatrixST.cs:294-319 Inverse (adjugate / det); returns -1 where the C# throws */
static int mst_inverse3(const mst *A, mst *Inv) {
    double det = mst_det3(A);
    if (!(det != 0)) return -1;
    mst_zero(Inv, 3, 3);
    double X = 1.0 / det;
    G(Inv, 0, 0) = X * (G(A, 1, 1) * G(A, 2, 2) - G(A, 1, 2) * G(A, 2, 1));
    G(Inv, 0, 1) = X * (G(A, 0, 2) * G(A, 2, 1) - G(A, 0, 1) * G(A, 2, 2));
    G(Inv, 0, 2) = X * (G(A, 0, 1) * G(A, 1, 2) - G(A, 0, 2) * G(A, 1, 1));
    G(Inv, 1, 0) = X * (G(A, 1, 2) * G(A, 2, 0) - G(A, 1, 0) * G(A, 2, 2));
    G(Inv, 1, 1) = X * (G(A, 0, 0) * G(A, 2, 2) - G(A, 0, 2) * G(A, 2, 0));
    G(Inv, 1, 2) = X * (G(A, 0, 2) * G(A, 1, 0) - G(A, 0, 0) * G(A, 1, 2));
    G(Inv, 2, 0) = X * (G(A, 1, 0) * G(A, 2, 1) - G(A, 1, 1) * G(A, 2, 0));
    G(Inv, 2, 1) = X * (G(A, 0, 1) * G(A, 2, 0) - G(A, 0, 0) * G(A, 2, 1));
    G(Inv, 2, 2) = X * (G(A, 0, 0) * G(A, 1, 1) - G(A, 0, 1) * G(A, 1, 0));
    return 0;
}
/* MatrixST.cs:327-339 MultiplyScalar */
static void mst_scale(const mst *A, double s, mst *C) {
    mst_zero(C, A->rows, A->cols);
    for (int i = 0; i < A->rows; i++)
        for (int j = 0; j < A->cols; j++) G(C, i, j) = G(A, i, j) * s;
}

/* ------------------------------------------------------------------------- */
/* FE_Library.cs:206-276 HEX8_Diff_ShapeFunctions, expressions kept verbatim   */
static void hex8_diff(double xi, double eta, double zeta, double *d /*3x8*/) {
    d[0 * 8 + 0] = 1.0 / 8.0 * (-1 + eta + zeta - eta * zeta);
    d[0 * 8 + 1] = 1.0 / 8.0 * (1 - eta - zeta + eta * zeta);
    d[0 * 8 + 2] = 1.0 / 8.0 * (1 + eta - zeta - eta * zeta);
    d[0 * 8 + 3] = 1.0 / 8.0 * (-1 - eta + zeta + eta * zeta);
    d[0 * 8 + 4] = 1.0 / 8.0 * (-1 + eta - zeta + eta * zeta);
    d[0 * 8 + 5] = 1.0 / 8.0 * (1 - eta + zeta - eta * zeta);
    d[0 * 8 + 6] = 1.0 / 8.0 * (1 + eta + zeta + eta * zeta);
    d[0 * 8 + 7] = 1.0 / 8.0 * (-1 - eta - zeta - eta * zeta);

    d[1 * 8 + 0] = 1.0 / 8.0 * (-1 + xi + zeta - xi * zeta);
    d[1 * 8 + 1] = 1.0 / 8.0 * (-1 - xi + zeta + xi * zeta);
    d[1 * 8 + 2] = 1.0 / 8.0 * (1 + xi - zeta - xi * zeta);
    d[1 * 8 + 3] = 1.0 / 8.0 * (1 - xi - zeta + xi * zeta);
    d[1 * 8 + 4] = 1.0 / 8.0 * (-1 + xi - zeta + xi * zeta);
    d[1 * 8 + 5] = 1.0 / 8.0 * (-1 - xi - zeta - xi * zeta);
    d[1 * 8 + 6] = 1.0 / 8.0 * (1 + xi + zeta + xi * zeta);
    d[1 * 8 + 7] = 1.0 / 8.0 * (1 - xi + zeta - xi * zeta);

    d[2 * 8 + 0] = 1.0 / 8.0 * (-1 + xi + eta - xi * eta);
    d[2 * 8 + 1] = 1.0 / 8.0 * (-1 - xi + eta + xi * eta);
    d[2 * 8 + 2] = 1.0 / 8.0 * (-1 - xi - eta - xi * eta);
    d[2 * 8 + 3] = 1.0 / 8.0 * (-1 + xi - eta + xi * eta);
    d[2 * 8 + 4] = 1.0 / 8.0 * (1 - xi - eta + xi * eta);
    d[2 * 8 + 5] = 1.0 / 8.0 * (1 + xi - eta - xi * eta);
    d[2 * 8 + 6] = 1.0 / 8.0 * (1 + xi + eta + xi * eta);
    d[2 * 8 + 7] = 1.0 / 8.0 * (1 - xi + eta - xi * eta);
}

/* Gauss point sign table, FE_Library.cs:121-128 (same order as the node table :108-115) */
static const int GPS[8][3] = {{-1, -1, -1}, {+1, -1, -1}, {+1, +1, -1}, {-1, +1, -1},
                              {-1, -1, +1}, {+1, -1, +1}, {+1, +1, +1}, {-1, +1, +1}};

static int type_ngp(int type) {
    if (type == STAN_HEX8_G1) return 1; /* FE_Library.cs:71 */
    if (type == STAN_HEX8_G2) return 8; /* FE_Library.cs:99 */
    return -1;
}
static double type_weight(int type) {
    return type == STAN_HEX8_G1 ? 2.0 * 2.0 * 2.0 /* :72 */ : 1.0 /* :100 */;
}

int stan_oracle_dn_dlocal(int type, int g, double out[24]) {
    int n = type_ngp(type);
    if (n < 0 || g < 0 || g >= n) return -1;
    if (type == STAN_HEX8_G1) {
        double GaussLocation = 0; /* FE_Library.cs:75 */
        hex8_diff(GaussLocation, GaussLocation, GaussLocation, out);
    } else {
        double GaussLocation = sqrt(1.0 / 3.0); /* FE_Library.cs:103 */
        hex8_diff(GPS[g][0] < 0 ? -GaussLocation : +GaussLocation,
                  GPS[g][1] < 0 ? -GaussLocation : +GaussLocation,
                  GPS[g][2] < 0 ? -GaussLocation : +GaussLocation, out);
    }
    return n;
}

/* FE_Library.cs:285-321 HEX8_ShapeFunctions(Node_Coord, GaussPointLoc) */
static void hex8_shape(const int nc[3], double loc, double n[8]) {
    double xi = nc[0] / loc, eta = nc[1] / loc, zeta = nc[2] / loc;
    n[0] = 1.0 / 8.0 * (1 - xi) * (1 - eta) * (1 - zeta);
    n[1] = 1.0 / 8.0 * (1 + xi) * (1 - eta) * (1 - zeta);
    n[2] = 1.0 / 8.0 * (1 + xi) * (1 + eta) * (1 - zeta);
    n[3] = 1.0 / 8.0 * (1 - xi) * (1 + eta) * (1 - zeta);
    n[4] = 1.0 / 8.0 * (1 - xi) * (1 - eta) * (1 + zeta);
    n[5] = 1.0 / 8.0 * (1 + xi) * (1 - eta) * (1 + zeta);
    n[6] = 1.0 / 8.0 * (1 + xi) * (1 + eta) * (1 + zeta);
    n[7] = 1.0 / 8.0 * (1 - xi) * (1 + eta) * (1 + zeta);
}

int stan_oracle_extrap_N(int type, double out[64]) {
    if (type == STAN_HEX8_G1) { /* FE_Library.cs:77-81: one row of ones */
        for (int i = 0; i < 8; i++) out[i] = 1.0;
        return 1;
    }
    if (type != STAN_HEX8_G2) return -1;
    double loc = sqrt(1.0 / 3.0);
    for (int i = 0; i < 8; i++) hex8_shape(GPS[i], loc, out + 8 * i); /* :105-116 */
    return 8;
}

/* Material.cs:31-56 */
void stan_oracle_material_D(double E, double Poisson, double D[36]) {
    memset(D, 0, 36 * sizeof(double));
    double lambda = (E * Poisson) / ((1 - 2 * Poisson) * (1 + Poisson));
    double Gm = (0.5 * E) / (1 + Poisson);
    D[0 * 6 + 0] = lambda + (2 * Gm);
    D[0 * 6 + 1] = lambda;
    D[0 * 6 + 2] = lambda;
    D[1 * 6 + 0] = lambda;
    D[1 * 6 + 1] = lambda + (2 * Gm);
    D[1 * 6 + 2] = lambda;
    D[2 * 6 + 0] = lambda;
    D[2 * 6 + 1] = lambda;
    D[2 * 6 + 2] = lambda + (2 * Gm);
    D[3 * 6 + 3] = Gm;
    D[4 * 6 + 4] = Gm;
    D[5 * 6 + 5] = Gm;
}

/* Element.cs:297-328 BL0_Matrix */
static void bl0_matrix(const mst *dN, mst *BL0) {
    mst_zero(BL0, 6, 24);
    for (int i = 0; i < 8; i++) {
        G(BL0, 0, 3 * i + 0) = G(dN, 0, i);
        G(BL0, 1, 3 * i + 1) = G(dN, 1, i);
        G(BL0, 2, 3 * i + 2) = G(dN, 2, i);
        G(BL0, 3, 3 * i + 0) = G(dN, 1, i);
        G(BL0, 3, 3 * i + 1) = G(dN, 0, i);
        G(BL0, 4, 3 * i + 1) = G(dN, 2, i);
        G(BL0, 4, 3 * i + 2) = G(dN, 1, i);
        G(BL0, 5, 3 * i + 0) = G(dN, 2, i);
        G(BL0, 5, 3 * i + 2) = G(dN, 0, i);
    }
}
/* Element.cs:333-366 BL1_Matrix (dU = Disp[inc], all zeros in linear statics) */
static void bl1_matrix(const mst *dN, const mst *dU, mst *BL1) {
    mst F;
    mst_mul(dN, dU, &F);
    mst_zero(BL1, 6, 24);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 3; j++) {
            G(BL1, 0, 3 * i + j) = G(&F, j, 0) * G(dN, 0, i);
            G(BL1, 1, 3 * i + j) = G(&F, j, 1) * G(dN, 1, i);
            G(BL1, 2, 3 * i + j) = G(&F, j, 2) * G(dN, 2, i);
            G(BL1, 3, 3 * i + j) = G(&F, j, 0) * G(dN, 1, i) + G(&F, j, 1) * G(dN, 0, i);
            G(BL1, 4, 3 * i + j) = G(&F, j, 1) * G(dN, 2, i) + G(&F, j, 2) * G(dN, 1, i);
            G(BL1, 5, 3 * i + j) = G(&F, j, 0) * G(dN, 2, i) + G(&F, j, 2) * G(dN, 0, i);
        }
}

/* per-Gauss-point pieces shared by K_Initial and Recovery_Stress:
 * J (Element.cs:274-292), dN (:130), BL = BL0 + BL1 (:135-143) */
static int gauss_point(const double xyz8[24], int type, int g, mst *J, mst *BL) {
    mst dNl, X, Inv, dN, U, BL0, BL1;
    mst_zero(&dNl, 3, 8);
    stan_oracle_dn_dlocal(type, g, dNl.m);
    mst_zero(&X, 8, 3);
    memcpy(X.m, xyz8, 24 * sizeof(double));
    mst_mul(&dNl, &X, J);                    /* Element.cs:289 */
    if (mst_inverse3(J, &Inv)) return -1;    /* MatrixST.cs:315-318 */
    mst_mul(&Inv, &dNl, &dN);                /* Element.cs:130 */
    mst_zero(&U, 8, 3);                      /* Element.cs:135-142: Disp[inc] == 0 (Node.cs:95-116) */
    bl0_matrix(&dN, &BL0);
    bl1_matrix(&dN, &U, &BL1);
    mst_add(&BL0, &BL1, BL);                 /* Element.cs:143 */
    return 0;
}

/* Element.cs:118-155 K_Initial */
int stan_oracle_ke_hex8(const double xyz8[24], const double D[36], int type, double K[576]) {
    int ngp = type_ngp(type);
    if (ngp < 0) return -2;
    double w = type_weight(type);
    mst Kacc, Dm;
    mst_zero(&Kacc, 24, 24);
    mst_zero(&Dm, 6, 6);
    memcpy(Dm.m, D, 36 * sizeof(double));
    for (int g = 0; g < ngp; g++) {
        mst J, BL, BLt, T, BDB, S, Ksum;
        if (gauss_point(xyz8, type, g, &J, &BL)) return -1;
        mst_transpose(&BL, &BLt);
        mst_mul(&BLt, &Dm, &T);   /* (BL^T * D) ...            Element.cs:151 */
        mst_mul(&T, &BL, &BDB);   /* ... * BL, left-associative               */
        mst_scale(&BDB, mst_det3(&J) * w, &S);
        mst_add(&Kacc, &S, &Ksum); /* K += ... == K = K + ...                 */
        Kacc = Ksum;
    }
    memcpy(K, Kacc.m, 576 * sizeof(double));
    return 0;
}

/* Element.cs:211-246 Recovery_Stress, :257-267 Update_StrainStress */
int stan_oracle_recover_hex8(const double xyz8[24], const double D[36], int type,
                             const double dU[24], double strain[48], double stress[48]) {
    if (type == STAN_HEX8_G1) return -4; /* N has one row; N[i][g] with i>=1 throws */
    int ngp = type_ngp(type);
    if (ngp < 0) return -2;
    double N[64];
    stan_oracle_extrap_N(type, N);
    double dE_g[8][6], dS_g[8][6];
    for (int g = 0; g < ngp; g++) {
        mst J, BL;
        if (gauss_point(xyz8, type, g, &J, &BL)) return -1;
        /* MatrixST.cs:347-367 MultiplyVector: C[i] += A[i,j]*V[j] */
        for (int i = 0; i < 6; i++) {
            double c = 0;
            for (int j = 0; j < 24; j++) c += G(&BL, i, j) * dU[j];
            dE_g[g][i] = c;
        }
        for (int i = 0; i < 6; i++) {
            double c = 0;
            for (int j = 0; j < 6; j++) c += D[i * 6 + j] * dE_g[g][j];
            dS_g[g][i] = c;
        }
    }
    /* Element.cs:238-245: dE[i] += row_g^T * N[i][g]; operator+ is 0 + (a + b) */
    for (int i = 0; i < 8; i++)
        for (int n = 0; n < 6; n++) {
            double e = 0, s = 0;
            for (int g = 0; g < ngp; g++) {
                e = e + dE_g[g][n] * N[i * 8 + g];
                s = s + dS_g[g][n] * N[i * 8 + g];
            }
            strain[i * 6 + n] = e;
            stress[i * 6 + n] = s;
        }
    return 0;
}

/* Element.cs:248-255 Compute_NodalForces, called right after Recovery_Stress (Solver.cs:186-187):
 *   NodalForces = sum_g (BL[g]^T * dS[g]) * (J[g].Det3() * GaussWeight)
 * where dS is the list filled by Recovery_Stress, i.e. the NODE-extrapolated stress increments,
 * indexed here by the Gauss point number (reference quirk, kept: SURVEY.md Appendix B).
 * stress_nodes = the [8][6] output of stan_oracle_recover_hex8. */
int stan_oracle_nodal_forces_hex8(const double xyz8[24], int type, const double stress_nodes[48],
                                  double forces[24]) {
    if (type == STAN_HEX8_G1) return -4; /* Recovery_Stress has thrown before this is reached */
    int ngp = type_ngp(type);
    if (ngp < 0) return -2;
    double w = type_weight(type);
    for (int r = 0; r < 24; r++) forces[r] = 0;
    for (int g = 0; g < ngp; g++) {
        mst J, BL;
        if (gauss_point(xyz8, type, g, &J, &BL)) return -1;
        double sc = mst_det3(&J) * w;
        for (int r = 0; r < 24; r++) { /* MatrixST.cs:404-418 operator*: inner index ascending */
            double c = 0;
            for (int k = 0; k < 6; k++) c += G(&BL, k, r) * stress_nodes[g * 6 + k];
            forces[r] = forces[r] + c * sc; /* operator+ allocates K + term (MatrixST.cs:327-339) */
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Database.cs:140-234 AssignDOF (literal)                                    */
int stan_oracle_assign_dof(int64_t n_nodes, int64_t n_elem, const int32_t *conn,
                           int32_t *node_index_out) {
    if (n_nodes <= 0) return -2;
    /* EList per node: Element.cs:474-480 AddElem2Nodes in ElemLib order, then
     * Node.cs:202-205 Distinct() (an element listing a node twice adds it twice) */
    int64_t *ecnt = calloc((size_t)n_nodes + 1, sizeof(int64_t));
    for (int64_t e = 0; e < n_elem; e++)
        for (int a = 0; a < 8; a++) ecnt[conn[e * 8 + a] + 1]++;
    for (int64_t i = 0; i < n_nodes; i++) ecnt[i + 1] += ecnt[i];
    int32_t *elist = malloc(sizeof(int32_t) * (size_t)(ecnt[n_nodes] ? ecnt[n_nodes] : 1));
    int64_t *fill = malloc(sizeof(int64_t) * (size_t)n_nodes);
    int32_t *ecount = calloc((size_t)n_nodes, sizeof(int32_t)); /* distinct count */
    for (int64_t i = 0; i < n_nodes; i++) fill[i] = ecnt[i];
    for (int64_t e = 0; e < n_elem; e++)
        for (int a = 0; a < 8; a++) {
            int32_t nd = conn[e * 8 + a];
            /* Distinct: skip if this element already recorded for nd (only the
             * previous entry can be the same element, entries are in element order) */
            if (fill[nd] > ecnt[nd] && elist[fill[nd] - 1] == (int32_t)e) continue;
            elist[fill[nd]++] = (int32_t)e;
            ecount[nd]++;
        }
    /* Neighbors: Database.cs:161-176 */
    int64_t *nptr = malloc(sizeof(int64_t) * ((size_t)n_nodes + 1));
    nptr[0] = 0;
    for (int64_t i = 0; i < n_nodes; i++) nptr[i + 1] = nptr[i] + 8 * (int64_t)ecount[i];
    int32_t *nbr = malloc(sizeof(int32_t) * (size_t)(nptr[n_nodes] ? nptr[n_nodes] : 1));
    int32_t *ncount = malloc(sizeof(int32_t) * (size_t)n_nodes);
    for (int64_t i = 0; i < n_nodes; i++) {
        int32_t *lst = nbr + nptr[i];
        int32_t cnt = 0;
        for (int32_t k = 0; k < ecount[i]; k++) {
            int32_t e = elist[ecnt[i] + k];
            for (int a = 0; a < 8; a++) {
                int32_t nid = conn[(int64_t)e * 8 + a];
                int dup = 0; /* Distinct(): keep first occurrence */
                for (int32_t q = 0; q < cnt; q++)
                    if (lst[q] == nid) { dup = 1; break; }
                if (!dup) lst[cnt++] = nid;
            }
        }
        /* N_neighbors.Remove(N.ID) */
        for (int32_t q = 0; q < cnt; q++)
            if (lst[q] == (int32_t)i) {
                memmove(lst + q, lst + q + 1, sizeof(int32_t) * (size_t)(cnt - q - 1));
                cnt--;
                break;
            }
        ncount[i] = cnt;
    }
    /* Find some peripheral Node: Database.cs:178-196 */
    int64_t first = -1;
    for (int c = 1; c < 7 && first < 0; c++)
        for (int64_t i = 0; i < n_nodes; i++)
            if (ecount[i] == c) { first = i; break; }
    int rc = 0;
    if (first < 0) { rc = -2; goto done; }
    {
        uint8_t *done_f = calloc((size_t)n_nodes, 1);
        /* NextNode = Neighbors[FirstNode] (aliased list that keeps growing) */
        int64_t cap = ncount[first] + 1024, len = ncount[first];
        int32_t *next = malloc(sizeof(int32_t) * (size_t)cap);
        memcpy(next, nbr + nptr[first], sizeof(int32_t) * (size_t)len);
        int32_t index = 0;
        node_index_out[first] = index++; /* SetDOF(index): DOF = 3*index + {0,1,2} */
        done_f[first] = 1;
        int64_t index2 = 0;
        while (index < n_nodes) {
            if (index2 >= len) { rc = -3; break; } /* NextNode[index2] out of range */
            int32_t nid = next[index2];
            if (!done_f[nid]) {
                node_index_out[nid] = index++;
                done_f[nid] = 1;
                for (int32_t q = 0; q < ncount[nid]; q++) {
                    int32_t n = nbr[nptr[nid] + q];
                    if (!done_f[n]) {
                        if (len == cap) {
                            cap *= 2;
                            next = realloc(next, sizeof(int32_t) * (size_t)cap);
                        }
                        next[len++] = n;
                    }
                }
            }
            index2++;
        }
        free(next);
        free(done_f);
    }
done:
    free(ncount);
    free(nbr);
    free(nptr);
    free(ecount);
    free(fill);
    free(elist);
    free(ecnt);
    return rc;
}

/* Solver.cs:121-132 */
int64_t stan_oracle_dof_reduction(int64_t n_dof, const uint8_t *fixed, int32_t *red) {
    int32_t reduc = 0;
    for (int64_t i = 0; i < n_dof; i++) {
        if (fixed[i]) {
            red[i] = -1;
            reduc++;
        } else
            red[i] = reduc;
    }
    return reduc;
}

/* ------------------------------------------------------------------------- */
/* ALGLIB hash-table sparse matrix (sparsecreate / sparseadd), open addressing */
typedef struct {
    int64_t n, cap, used;
    int64_t *key; /* i*n + j, -1 = free */
    double *val;
} shash;

static uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
static void shash_init(shash *h, int64_t n, int64_t cap) {
    h->n = n;
    h->cap = cap;
    h->used = 0;
    h->key = malloc(sizeof(int64_t) * (size_t)cap);
    h->val = malloc(sizeof(double) * (size_t)cap);
    for (int64_t i = 0; i < cap; i++) h->key[i] = -1;
}
static void shash_add(shash *h, int64_t k, double v);
static void shash_grow(shash *h) {
    shash g;
    shash_init(&g, h->n, h->cap * 2);
    for (int64_t i = 0; i < h->cap; i++)
        if (h->key[i] >= 0) {
            uint64_t p = mix64((uint64_t)h->key[i]) & (uint64_t)(g.cap - 1);
            while (g.key[p] >= 0) p = (p + 1) & (uint64_t)(g.cap - 1);
            g.key[p] = h->key[i];
            g.val[p] = h->val[i];
            g.used++;
        }
    free(h->key);
    free(h->val);
    *h = g;
}
/* sparseadd: S[i,j] += v, creating the entry when absent */
static void shash_add(shash *h, int64_t k, double v) {
    if (2 * (h->used + 1) > h->cap) shash_grow(h);
    uint64_t p = mix64((uint64_t)k) & (uint64_t)(h->cap - 1);
    while (h->key[p] >= 0 && h->key[p] != k) p = (p + 1) & (uint64_t)(h->cap - 1);
    if (h->key[p] < 0) {
        h->key[p] = k;
        h->val[p] = v; /* new entry starts from v (0 + v) */
        h->used++;
    } else
        h->val[p] += v;
}

static int cmp_i32(const void *a, const void *b) {
    int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

/* sparseconverttocrs: rows ascending, columns ascending within a row */
static void shash_to_crs(shash *h, stan_oracle_crs *m) {
    int64_t n = h->n;
    m->n = n;
    m->nnz = h->used;
    m->ridx = calloc((size_t)n + 1, sizeof(int64_t));
    m->idx = malloc(sizeof(int32_t) * (size_t)(h->used ? h->used : 1));
    m->vals = malloc(sizeof(double) * (size_t)(h->used ? h->used : 1));
    for (int64_t i = 0; i < h->cap; i++)
        if (h->key[i] >= 0) m->ridx[h->key[i] / n + 1]++;
    for (int64_t i = 0; i < n; i++) m->ridx[i + 1] += m->ridx[i];
    int64_t *fill = malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
    memcpy(fill, m->ridx, sizeof(int64_t) * (size_t)n);
    for (int64_t i = 0; i < h->cap; i++)
        if (h->key[i] >= 0) m->idx[fill[h->key[i] / n]++] = (int32_t)(h->key[i] % n);
    for (int64_t r = 0; r < n; r++)
        qsort(m->idx + m->ridx[r], (size_t)(m->ridx[r + 1] - m->ridx[r]), sizeof(int32_t), cmp_i32);
    /* values: look each (r,c) up again */
    for (int64_t r = 0; r < n; r++)
        for (int64_t q = m->ridx[r]; q < m->ridx[r + 1]; q++) {
            int64_t k = r * n + m->idx[q];
            uint64_t p = mix64((uint64_t)k) & (uint64_t)(h->cap - 1);
            while (h->key[p] != k) p = (p + 1) & (uint64_t)(h->cap - 1);
            m->vals[q] = h->val[p];
        }
    free(fill);
}

void stan_oracle_crs_free(stan_oracle_crs *m) {
    free(m->ridx);
    free(m->idx);
    free(m->vals);
    memset(m, 0, sizeof(*m));
}

/* SolverFunctions.cs:117-180 ParallelAssembly_K */
int stan_oracle_assemble(int64_t n_nodes, const double *xyz, const int32_t *node_dof,
                         int64_t n_elem, const int32_t *conn, const int32_t *elem_mat,
                         const uint8_t *elem_type, int32_t n_mat, const double *mat_E_nu,
                         int64_t n_dof, const int32_t *red, int n_threads,
                         stan_oracle_crs *out, int64_t *bad_elem) {
    (void)n_nodes;
    int64_t nfix = 0;
    for (int64_t i = 0; i < n_dof; i++) nfix += (red[i] == -1); /* :122 */
    int64_t N = n_dof - nfix;
    double *Dm = malloc(sizeof(double) * 36 * (size_t)(n_mat > 0 ? n_mat : 1));
    for (int32_t m = 0; m < n_mat; m++)
        stan_oracle_material_D(mat_E_nu[2 * m], mat_E_nu[2 * m + 1], Dm + 36 * m);
    shash h;
    int64_t cap = 1024;
    while (cap < 64) cap *= 2;
    shash_init(&h, N > 0 ? N : 1, cap); /* :123 sparsecreate */
    int rc = 0;
    const int64_t CH = 4096; /* K_e computed chunk-wise in parallel, scattered in element order */
    double *kbuf = malloc(sizeof(double) * 576 * (size_t)CH);
    int *krc = malloc(sizeof(int) * (size_t)CH);
    if (n_threads < 1) n_threads = 1;
    for (int64_t e0 = 0; e0 < n_elem && rc == 0; e0 += CH) {
        int64_t e1 = e0 + CH < n_elem ? e0 + CH : n_elem;
#ifdef _OPENMP
#pragma omp parallel for num_threads(n_threads) schedule(static)
#endif
        for (int64_t e = e0; e < e1; e++) {
            double x8[24];
            for (int a = 0; a < 8; a++) {
                int32_t nd = conn[e * 8 + a];
                x8[3 * a + 0] = xyz[3 * (int64_t)nd + 0];
                x8[3 * a + 1] = xyz[3 * (int64_t)nd + 1];
                x8[3 * a + 2] = xyz[3 * (int64_t)nd + 2];
            }
            krc[e - e0] = stan_oracle_ke_hex8(x8, Dm + 36 * elem_mat[e], elem_type[e],
                                              kbuf + 576 * (e - e0)); /* :135 */
        }
        for (int64_t e = e0; e < e1; e++) {
            if (krc[e - e0]) {
                rc = krc[e - e0];
                if (bad_elem) *bad_elem = e;
                break;
            }
            const double *k = kbuf + 576 * (e - e0);
            /* :143-173 */
            for (int i = 0; i < 8; i++)
                for (int m = 0; m < 3; m++)
                    for (int j = 0; j < 8; j++)
                        for (int n = 0; n < 3; n++) {
                            int32_t row = node_dof[3 * (int64_t)conn[e * 8 + i] + m];
                            int32_t col = node_dof[3 * (int64_t)conn[e * 8 + j] + n];
                            if (col >= row && red[row] != -1 && red[col] != -1)
                                shash_add(&h, (int64_t)(row - red[row]) * h.n + (col - red[col]),
                                          k[(i * 3 + m) * 24 + (j * 3 + n)]);
                        }
        }
    }
    free(kbuf);
    free(krc);
    free(Dm);
    if (rc == 0) {
        shash_to_crs(&h, out); /* SolverFunctions.cs:275 sparseconverttocrs */
        out->n = N;
        if (N == 0) out->nnz = 0;
    }
    free(h.key);
    free(h.val);
    return rc;
}

/* ALGLIB sparsesmv(S, isupper=true, x, y): y = S*x with S symmetric, given by
 * its upper triangle (diagonal + strictly-upper part applied both ways). */
void stan_oracle_smv_upper(const stan_oracle_crs *A, const double *x, double *y) {
    int64_t n = A->n;
    for (int64_t i = 0; i < n; i++) y[i] = 0;
    for (int64_t i = 0; i < n; i++) {
        int64_t q = A->ridx[i], q1 = A->ridx[i + 1];
        if (q < q1 && A->idx[q] == i) {
            y[i] += A->vals[q] * x[i];
            q++;
        }
        double vx = x[i], vy = 0;
        for (; q < q1; q++) {
            int32_t j = A->idx[q];
            double v = A->vals[q];
            y[j] += vx * v;
            vy += x[j] * v;
        }
        y[i] += vy;
    }
}

/* ---- optional all-cores mode of the CG's matrix-vector product (NOT what the reference does:
 * alglib's sparsesmv is serial; this only provides the second, clearly labelled CPU number of
 * BASELINE.md section 2).  The symmetric matrix is expanded to full CRS once and rows are
 * gathered in parallel. */
static int g_mv_threads = 1;
void stan_oracle_set_mv_threads(int n) { g_mv_threads = n > 1 ? n : 1; }

typedef struct { int64_t *rp; int32_t *ci; double *v; } full_crs;
static void full_from_upper(const stan_oracle_crs *A, full_crs *F) {
    int64_t n = A->n;
    F->rp = calloc((size_t)n + 1, sizeof(int64_t));
    for (int64_t i = 0; i < n; i++)
        for (int64_t q = A->ridx[i]; q < A->ridx[i + 1]; q++) {
            F->rp[i + 1]++;
            if (A->idx[q] != i) F->rp[(int64_t)A->idx[q] + 1]++;
        }
    for (int64_t i = 0; i < n; i++) F->rp[i + 1] += F->rp[i];
    F->ci = malloc(sizeof(int32_t) * (size_t)(F->rp[n] ? F->rp[n] : 1));
    F->v = malloc(sizeof(double) * (size_t)(F->rp[n] ? F->rp[n] : 1));
    int64_t *fill = malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
    memcpy(fill, F->rp, sizeof(int64_t) * (size_t)n);
    /* pass 1: transposed strictly-upper entries (row order => ascending columns), pass 2: upper */
    for (int64_t i = 0; i < n; i++)
        for (int64_t q = A->ridx[i]; q < A->ridx[i + 1]; q++)
            if (A->idx[q] != i) {
                int64_t j = A->idx[q];
                F->ci[fill[j]] = (int32_t)i;
                F->v[fill[j]++] = A->vals[q];
            }
    for (int64_t i = 0; i < n; i++)
        for (int64_t q = A->ridx[i]; q < A->ridx[i + 1]; q++) {
            F->ci[fill[i]] = A->idx[q];
            F->v[fill[i]++] = A->vals[q];
        }
    free(fill);
}
static void full_mv(const full_crs *F, int64_t n, const double *x, double *y) {
#ifdef _OPENMP
#pragma omp parallel for num_threads(g_mv_threads) schedule(static)
#endif
    for (int64_t i = 0; i < n; i++) {
        double a = 0;
        for (int64_t q = F->rp[i]; q < F->rp[i + 1]; q++) a += F->v[q] * x[F->ci[q]];
        y[i] = a;
    }
}

/* SolverFunctions.cs:270-330 -> alglib.lincg* (3.16.0), restated from the
 * published algorithm: diagonal preconditioner applied as symmetric scaling,
 * x0 = 0, residual refresh every 10 iterations, merit-function stop. */
int stan_oracle_cg(const stan_oracle_crs *A, const double *b_in, double epsf, int32_t maxits,
                   double *x_out, int32_t *terminationtype, int32_t *iterations, int32_t *nmv_out,
                   double *rel_res) {
    return stan_oracle_cg_opt(A, b_in, epsf, maxits, 1, 10, x_out, terminationtype, iterations,
                              nmv_out, rel_res);
}

/* merit_stop = 0 / itsbeforerupdate != 10 are NOT ALGLIB defaults: they mirror the
 * STAN_OPT_* switches of libstan_hip.so so that both sides can run the same variant. */
int stan_oracle_cg_opt(const stan_oracle_crs *A, const double *b_in, double epsf, int32_t maxits,
                       int merit_stop, int itsbeforerupdate, double *x_out,
                       int32_t *terminationtype, int32_t *iterations, int32_t *nmv_out,
                       double *rel_res) {
    int64_t n = A->n;
    const int itsbeforerestart = (int)(n > 0x7fffffff ? 0x7fffffff : n); /* lincgcreate */
    if (epsf == 0 && maxits == 0) epsf = 1.0e-6; /* lincgsetcond */
    size_t sz = sizeof(double) * (size_t)(n ? n : 1);
    double *s = malloc(sz), *b = malloc(sz), *rx = malloc(sz), *cx = malloc(sz), *r = malloc(sz),
           *cr = malloc(sz), *p = malloc(sz), *z = malloc(sz), *cz = malloc(sz), *mv = malloc(sz),
           *t = malloc(sz);
    /* lincgsolvesparse: s_i = 1/sqrt(A_ii) if A_ii > 0 else 1 */
    for (int64_t i = 0; i < n; i++) {
        double v = 0;
        if (A->ridx[i] < A->ridx[i + 1] && A->idx[A->ridx[i]] == i) v = A->vals[A->ridx[i]];
        s[i] = v > 0 ? 1 / sqrt(v) : 1;
    }
    for (int64_t i = 0; i < n; i++) b[i] = b_in[i] * s[i];
    int nmv = 0, its = 0, term = 0;
    const int use_full = g_mv_threads > 1;
    full_crs Ffull = {0, 0, 0};
    if (use_full) full_from_upper(A, &Ffull);
#define MV(vec, vmv)                                         \
    do {                                                     \
        for (int64_t i_ = 0; i_ < n; i_++) t[i_] = (vec)[i_] * s[i_]; \
        if (use_full) full_mv(&Ffull, n, t, mv);             \
        else stan_oracle_smv_upper(A, t, mv);                \
        (vmv) = 0;                                           \
        for (int64_t i_ = 0; i_ < n; i_++) {                 \
            mv[i_] *= s[i_];                                 \
            (vmv) += (vec)[i_] * mv[i_];                     \
        }                                                    \
        nmv++;                                               \
    } while (0)
    double vmv, bnorm = 0, r2 = 0, merit = 0, prevmf;
    for (int64_t i = 0; i < n; i++) rx[i] = 0; /* startx = 0 */
    MV(rx, vmv);
    for (int64_t i = 0; i < n; i++) {
        r[i] = b[i] - mv[i];
        r2 += r[i] * r[i];
        merit += mv[i] * rx[i] - 2 * b[i] * rx[i];
        bnorm += b[i] * b[i];
    }
    prevmf = merit;
    bnorm = sqrt(bnorm);
    if (!isfinite(r2)) {
        term = -4;
        goto finish;
    }
    if (sqrt(r2) <= epsf * bnorm) {
        term = 1;
        goto finish;
    }
    for (int64_t i = 0; i < n; i++) { /* unit preconditioner on the scaled system */
        z[i] = r[i];
        p[i] = z[i];
    }
    for (;;) {
        its++;
        MV(p, vmv);
        if (!isfinite(vmv) || vmv <= 0) {
            term = isfinite(vmv) ? -5 : -4;
            break;
        }
        double alpha = 0;
        for (int64_t i = 0; i < n; i++) alpha += r[i] * z[i];
        alpha /= vmv;
        if (!isfinite(alpha)) {
            term = -4;
            break;
        }
        for (int64_t i = 0; i < n; i++) cx[i] = rx[i] + alpha * p[i];
        merit = 0;
        if (!(itsbeforerupdate == 0 || its % itsbeforerupdate != 0)) {
            double dummy;
            MV(cx, dummy);
            (void)dummy;
            for (int64_t i = 0; i < n; i++) {
                cr[i] = b[i] - mv[i];
                merit += (mv[i] - 2 * b[i]) * cx[i];
            }
        } else {
            for (int64_t i = 0; i < n; i++) {
                cr[i] = r[i] - alpha * mv[i];
                merit -= (cr[i] + b[i]) * cx[i];
            }
        }
        r2 = 0;
        for (int64_t i = 0; i < n; i++) r2 += cr[i] * cr[i];
        if (sqrt(r2) <= epsf * bnorm) {
            memcpy(rx, cx, sz);
            term = 1;
            break;
        }
        if (its >= maxits && maxits > 0) {
            memcpy(rx, cx, sz);
            term = 5;
            break;
        }
        if (merit_stop && merit >= prevmf) { /* no further progress: keep the previous (best) point */
            term = 7;
            for (int64_t i = 0; i < n; i++)
                if (!isfinite(rx[i])) term = -4;
            break;
        }
        memcpy(rx, cx, sz);
        prevmf = merit;
        for (int64_t i = 0; i < n; i++) cz[i] = cr[i];
        if (its % itsbeforerestart != 0) {
            double beta = 0, uvar = 0;
            for (int64_t i = 0; i < n; i++) {
                beta += cz[i] * cr[i];
                uvar += z[i] * r[i];
            }
            beta /= uvar;
            if (!isfinite(beta)) {
                term = -4;
                break;
            }
            for (int64_t i = 0; i < n; i++) p[i] = cz[i] + beta * p[i];
        } else
            for (int64_t i = 0; i < n; i++) p[i] = cz[i];
        memcpy(r, cr, sz);
        memcpy(z, cz, sz);
    }
finish:
#undef MV
    for (int64_t i = 0; i < n; i++) x_out[i] = rx[i] * s[i];
    if (terminationtype) *terminationtype = term;
    if (iterations) *iterations = its;
    if (nmv_out) *nmv_out = nmv;
    if (rel_res) *rel_res = bnorm > 0 ? sqrt(r2) / bnorm : 0;
    free(s); free(b); free(rx); free(cx); free(r); free(cr); free(p); free(z); free(cz);
    free(mv); free(t);
    if (use_full) { free(Ffull.rp); free(Ffull.ci); free(Ffull.v); }
    return 0;
}

/* SolverFunctions.cs:520-538 */
void stan_oracle_include_bc(int64_t n_dof, const int32_t *red, const double *U, double *U_full) {
    for (int64_t i = 0; i < n_dof; i++) U_full[i] = red[i] == -1 ? 0 : U[i - red[i]];
}
