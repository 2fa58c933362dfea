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

#ifdef __cplusplus
}
#endif
#endif
