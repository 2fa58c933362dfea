/*
 * stan_host.h -- C-ABI of libstan_host.so: the host-side (integer / bookkeeping) steps that
 * sit directly before and after the GPU hot path in Solver.SolverLinearStatics
 * (Solver.cs:71-217), plus the STdb database codec.  Pure C++17, no GPU, no HIP: it loads
 * and runs on any host.  The C++ object model behind it (stan_amd/host/model.h) mirrors
 * STAN_Database's classes (Database, Node, Element, Material, BoundaryCondition, Analysis).
 *
 * All functions return 0 on success or a negative STAN_HOST_E_* code.
 */
#ifndef STAN_HOST_H
#define STAN_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STAN_HOST_OK 0
#define STAN_HOST_E_ARG (-2)
#define STAN_HOST_E_NO_START (-20)     /* Database.cs:179-196 finds no node with 1..6 elements   */
#define STAN_HOST_E_DISCONNECTED (-21) /* Database.cs:218 NextNode[index2] out of range          */
#define STAN_HOST_E_IO (-22)
#define STAN_HOST_E_FORMAT (-23)
#define STAN_HOST_E_MEMORY (-24)       /* the skyline profile does not fit (direct solver)       */

/* Database.AssignDOF (Database.cs:140-234): BFS node numbering, bit-exact with the
 * reference's neighbour order (element iteration order x NList order, first occurrence).
 * conn [n_elem*8]: node *indices* in NodeLib order.  node_index_out[i] = BFS index, so
 * Node.DOF = {3*idx, 3*idx+1, 3*idx+2} (Node.cs:218-223) is written to node_dof_out
 * [n_nodes*3] when non-NULL. */
int stan_host_assign_dof(int64_t n_nodes, int64_t n_elem, const int32_t *conn,
                         int32_t *node_index_out, int32_t *node_dof_out);

/* Solver.cs:104-132: Fix_DOF from SPC entries whose value == 1 exactly, then
 * nDOF_reduction.  spc_nodes [n_spc] node indices, spc_vals [n_spc*3] the 3x1 NodalValues.
 * red_out [n_dof].  *n_fixed_out = number of distinct fixed DOFs. */
int stan_host_dof_reduction(int64_t n_dof, const int32_t *node_dof, int64_t n_spc,
                            const int32_t *spc_nodes, const double *spc_vals, int32_t *red_out,
                            int64_t *n_fixed_out);

/* Solver.cs:136-152: F[dof - red[dof]] += value on free DOFs (duplicates accumulate).
 * F_out [n_dof - n_fixed] must be zero-initialised by the caller. */
int stan_host_load_vector(int64_t n_dof, const int32_t *node_dof, const int32_t *red,
                          int64_t n_load, const int32_t *load_nodes, const double *load_vals,
                          double *F_out);

/* SolverFunctions.cs:520-538 Include_BC_DOF + Solver.cs:171-178 write-back:
 * disp_out[i*3+j] = U_full[node_dof[i*3+j]], U_full = 0 on fixed DOFs. */
int stan_host_nodal_displacements(int64_t n_nodes, const int32_t *node_dof, const int32_t *red,
                                  const double *U, double *disp_out);

/* ---- direct solvers: CPU fallback for Analysis.LinSolver = "Cholesky" / "LU" -----------------
 * (SolverFunctions.cs:332-444, 446-516; not on the GPU hot path).  K = the reduced upper-triangle
 * CRS the reference's alglib.sparsematrix holds: rowptr [n+1], col/val [nnz], col >= row
 * (stan_hip_matrix_to_csr with upper_only = 1).  b, x [n].
 * cholesky: in-place skyline A = U^T U without a profile-reducing permutation, like
 * alglib.sparsecholeskyskyline; *termination_type = 1 and x = solution, or -3 and x = 0 when the
 * matrix is not positive definite (sparsecholeskysolvesks' report).  *profile_entries (may be NULL)
 * = doubles in the skyline; more than 2^32 of them is STAN_HOST_E_MEMORY.
 * lu: what the reference's LinearSolver_LU returns -- alglib.sparselu is handed the stored UPPER
 * triangle as a general matrix, so x solves triu(K) x = b (a reference quirk, kept). */
int stan_host_cholesky_skyline_solve(int64_t n, const int64_t *rowptr, const int32_t *col,
                                     const double *val, const double *b, double *x,
                                     int32_t *termination_type, int64_t *profile_entries);
int stan_host_lu_upper_solve(int64_t n, const int64_t *rowptr, const int32_t *col, const double *val,
                             const double *b, double *x, int32_t *termination_type);

/* ---- row partition + halo plan of the sharded CG (no counterpart in the reference) -----------
 * Block rows = nodes in reference DOF order.  row_starts [nranks+1]. */
int stan_host_partition_rows(int64_t n_block_rows, int32_t nranks, int64_t *row_starts);
/* Plan of `rank`: halo_glob [<= n_nodes] ascending global block rows it must receive;
 * neighbours nbr_ranks [<= nranks] ascending; send_rows [<= n_nodes] LOCAL owned rows grouped by
 * neighbour, send_off / recv_off [n_nbr+1] offsets into send_rows / halo_glob.  Output arrays
 * other than row_starts, n_halo, n_nbr may be NULL (sizes only). */
int stan_host_partition_plan(int64_t n_nodes, const int32_t *node_index, int64_t n_elem,
                             const int32_t *conn, int32_t nranks, int32_t rank, int64_t *row_starts,
                             int64_t *n_halo, int32_t *halo_glob, int32_t *n_nbr,
                             int32_t *nbr_ranks, int64_t *send_off, int32_t *send_rows,
                             int64_t *recv_off);

/* Elements `rank` has to hold: every element with a node in its block-row range (boundary elements
 * go to both owners).  elem_idx_out [<= n_elem] ascending; a launcher that keeps its inputs on the
 * device passes exactly these to stan_hip_assemble_hex8_dev (the host-pointer entry filters itself). */
int stan_host_partition_elements(int64_t n_nodes, const int32_t *node_index, int64_t n_elem,
                                 const int32_t *conn, int32_t nranks, int32_t rank,
                                 int32_t *elem_idx_out, int64_t *n_out);

/* ---- STAN_Database object model + STdb codec (stan_amd/host/model.h) -----------------------
 * stan_db wraps a Database (Database.cs:10-21).  Text comes back through
 * stan_host_db_last_error.  Strings are UTF-8, NUL-terminated. */
typedef struct stan_db stan_db;
int stan_host_db_new(stan_db **out);
void stan_host_db_free(stan_db *db);
const char *stan_host_db_last_error(stan_db *db);

/* ProtoDeserialize / ProtoSerialize (SolverFunctions.cs:48-63).  packed != 0 writes repeated
 * scalars packed; the reader accepts both. */
int stan_host_db_read_stdb(stan_db *db, const char *path);
int stan_host_db_parse_stdb(stan_db *db, const uint8_t *data, int64_t size);
int stan_host_db_write_stdb(stan_db *db, const char *path, int32_t packed);
/* ExportOutput after a solve (Solver.cs:454-462) WITHOUT storing the results in the objects first: the file
 * is byte for byte what stan_host_db_set_results + stan_host_db_write_stdb write (node i: DispX/Y/Z = {0,
 * disp[3i+d]}; element e: Strain / Stress = {zeros(8x6), strain[48e..] / stress[48e..]}; Result_StepNo = 1),
 * encoded straight from the flat arrays (disp [n_nodes*3] NodeLib order, strain / stress [n_elem*48]).  The
 * database itself is left unchanged. */
int stan_host_db_write_stdb_with_results(stan_db *db, const char *path, int32_t packed, const double *disp,
                                         const double *strain, const double *stress);
/* two-call: buf == NULL returns the size */
int stan_host_db_serialize(stan_db *db, int32_t packed, uint8_t *buf, int64_t cap, int64_t *size);

/* Database.ReadNastranMesh (Database.cs:39-111) + Set_nDOF (:135-138), what the GUI's
 * Import does (MainWindow.xaml.cs:181-238).  *n_import_errors = lines that failed to parse. */
int stan_host_db_read_bdf(stan_db *db, const char *path, int64_t *n_import_errors);
/* Bulk mesh construction from arrays (IDs, not indices); Type of every element = hex_type. */
int stan_host_db_set_mesh(stan_db *db, int64_t n_nodes, const int32_t *node_ids, const double *xyz,
                          int64_t n_elem, const int32_t *elem_ids, const int32_t *elem_pids,
                          const int32_t *nlist8, const char *hex_type);
/* new Material(id) + SetElastic / Name (Material.cs:19-56) */
int stan_host_db_add_material(stan_db *db, int32_t id, const char *name, double E, double nu);
/* What the GUI's part box does before saving: MatID and FE type of every element of a
 * part (Part.cs:658-673, 767-774) and the PartInfo record (MainWindow.xaml.cs:456-462). */
int stan_host_db_assign_part(stan_db *db, int32_t pid, int32_t mat_id, const char *hex_type);
/* new BoundaryCondition(name, type, id) + Add per node (BoundaryCondition.cs:29-37, 87-98):
 * type "SPC" or "PointLoad"; vals [n*3]; unknown node IDs are ignored like the reference. */
int stan_host_db_add_bc(stan_db *db, int32_t id, const char *name, const char *type, int64_t n,
                        const int32_t *node_ids, const double *vals);
int stan_host_db_set_analysis(stan_db *db, const char *type, const char *lin_solver, double tol,
                              int32_t max_iter, int32_t inc_numb);

/* sizes[0..7] = nodes, elements, materials, BCs, nDOF, Result_StepNo, import errors, 0 */
int stan_host_db_sizes(stan_db *db, int64_t sizes[8]);
/* analysis settings: type/lin_solver copied into caller buffers of cap bytes */
int stan_host_db_get_analysis(stan_db *db, char *type, char *lin_solver, int32_t cap, double *tol,
                              int32_t *max_iter, int32_t *result_step);
int stan_host_db_assign_dof(stan_db *db); /* Database.AssignDOF, Solver.cs:46 */
/* Flat arrays for stan_hip_assemble_hex8 (any pointer may be NULL to skip it):
 * xyz[n*3], node_ids[n], node_dof[n*3], conn[e*8] (indices), elem_ids[e], elem_mat[e],
 * elem_type[e], mat_E_nu[cap_mat*2]; *n_mat = materials used. */
int stan_host_db_get_flat(stan_db *db, double *xyz, int32_t *node_ids, int32_t *node_dof,
                          int32_t *conn, int32_t *elem_ids, int32_t *elem_mat,
                          uint8_t *elem_type, double *mat_E_nu, int32_t cap_mat, int32_t *n_mat);
/* Solver.cs:104-152 from the BCLib: red[nDOF], *n_fixed, F[nDOF - n_fixed] (F may be NULL
 * on a first call that only asks for n_fixed). */
int stan_host_db_get_reduction(stan_db *db, int32_t *red, int64_t *n_fixed, double *F);
/* Result write-back of SolverLinearStatics (Solver.cs:81-90, 171-178, 203-210, Main :56):
 * re-initialises step 0/1, stores disp[n*3] into Node.DispX/Y/Z[1] and strain/stress
 * [e*48] (may be NULL) into Element.Strain[1]/Stress[1], sets Result_StepNo = 1. */
int stan_host_db_set_results(stan_db *db, const double *disp, const double *strain,
                             const double *stress);
int stan_host_db_get_results(stan_db *db, int32_t inc, double *disp, double *strain,
                             double *stress);

#ifdef __cplusplus
}
#endif
#endif
