// StanHip.cs -- P/Invoke surface of libstan_hip.so (include/stan_hip.h), one declaration per exported function.
// New file for src/STAN_Solver/ of galuszkm/STAN; nothing in it depends on the rest of the solver.
//
// Kept in step with the header MECHANICALLY: tests/test_integration_shim.py parses every [DllImport] below and
// every prototype of include/stan_hip.h and fails on a missing function, a different arity, or an argument whose
// managed type is not the blittable image of the C type (int32_t -> int, int64_t -> long, double -> double,
// T* -> T[] / out T / IntPtr, handles -> IntPtr, T** -> out IntPtr); the two [StructLayout] records are checked
// field by field.  No .NET toolchain exists in the build image: this file is checked by that test, not by csc.
using System;
using System.Runtime.InteropServices;

namespace STAN_Solver
{
    /// stan_matrix_info (include/stan_hip.h)
    [StructLayout(LayoutKind.Sequential)]
    public struct StanMatrixInfo
    {
        public long n_dof;
        public long n_reduced;
        public long n_block_rows;
        public long row_begin;
        public long row_end;
        public long n_halo;
        public long n_blocks;
        public long n_slots;
        public long bytes_matrix;
        public int scaled;
        public int max_row_blocks;
        public long n_elements_on_device;
        public int sell_sigma;
        public int folded_slots_permille;
    }

    /// stan_profile (include/stan_hip.h)
    [StructLayout(LayoutKind.Sequential)]
    public struct StanProfile
    {
        public double assemble_ms;
        public double symbolic_ms;
        public double numeric_ms;
        public double cg_ms;
        public double spmv_ms_total;
        public long spmv_launches;
        public long spmv_bytes;
        public long cg_iteration_vector_bytes;
        public int iterations;
        public int termination_type;
        public int assembly_colours;
        public int value_stream;
        public double spmv2_ms_total;
        public long spmv2_launches;
        public long loop_kernel_launches;
        public long loop_collectives;
        public long loop_iterations_enqueued;
        public int placement_candidates;
        public float placement_ms_best;
        public float placement_ms_worst;
        public long col_slots_packed;
        public int placement_moved_vectors;
        public int repacked_streams;
        public long loop_stream_waits;
        public double comm_reduce_ms_total;
        public long comm_reduce_calls;
        public double comm_halo_ms_total;
        public long comm_halo_calls;
        public double rel_residual_recurrence;
        public double rel_residual_fp64;
        public int refine_passes;
        public int fp64_products;
        public double fp64_products_ms;
    }

    internal static class StanHipNative
    {
        // "stan_hip" resolves to libstan_hip.so on Linux (.NET probes lib<name>.so; Mono: <dllmap> or the same probing)
        // and to stan_hip.dll on Windows.  The directory must be on LD_LIBRARY_PATH / next to the executable.
        const string Lib = "stan_hip";

        // error codes, element types, precision modes, options (the #defines of the header)
        internal const int STAN_OK = 0, STAN_E_HIP = -1, STAN_E_ARG = -2, STAN_E_ALLOC = -3, STAN_E_DETJ = -4,
                           STAN_E_DOF_LAYOUT = -5, STAN_E_VALENCE = -6, STAN_E_COMM = -7, STAN_E_UNSUPPORTED = -8;
        internal const byte STAN_HEX8_G1 = 1, STAN_HEX8_G2 = 2;
        internal const int STAN_PREC_FP64 = 0, STAN_PREC_MIXED = 1, STAN_PREC_FIXED48 = 2;
        internal const int STAN_OPT_CG_MERIT_STOP = 1, STAN_OPT_CG_RUPDATE = 2, STAN_OPT_SPMV_VARIANT = 3,
                           STAN_OPT_OVERLAP_HALO = 4, STAN_OPT_ASSEMBLY_MODE = 5, STAN_OPT_CG_FUSED_REFRESH = 6,
                           STAN_OPT_POOL = 7, STAN_OPT_PLACEMENT_TRIES = 8, STAN_OPT_POOL_MAX_BYTES = 9,
                           STAN_OPT_CG_SINGLE_REDUCE = 10, STAN_OPT_CG_FOLD_REDUCE = 11, STAN_OPT_VEC_STORE_NT = 12,
                           STAN_OPT_PACKED_COLUMNS = 13, STAN_OPT_CG_DEFER_X = 14, STAN_OPT_SPMV_SMALL = 15,
                           STAN_OPT_PLACEMENT_MAX_BYTES = 16, STAN_OPT_SELL_SIGMA = 17, STAN_OPT_COMM_P2P = 18,
                           STAN_OPT_ROW_FOLDING = 19, STAN_OPT_CG_REFINE = 20, STAN_OPT_CG_LAZY_SCALING = 21;

        // ---- context
        [DllImport(Lib)] internal static extern int stan_hip_init(int device, out IntPtr ctx);
        [DllImport(Lib)] internal static extern int stan_hip_init_multi(int n_devices, int[] devices, out IntPtr ctx);
        [DllImport(Lib)] internal static extern void stan_hip_destroy(IntPtr ctx);
        [DllImport(Lib)] internal static extern IntPtr stan_hip_last_error(IntPtr ctx);
        [DllImport(Lib)] internal static extern long stan_hip_last_bad_element(IntPtr ctx);
        [DllImport(Lib)] internal static extern int stan_hip_set_stream(IntPtr ctx, IntPtr hip_stream);
        [DllImport(Lib)] internal static extern int stan_hip_set_option(IntPtr ctx, int option, long value);
        [DllImport(Lib)] internal static extern int stan_hip_pool_info(IntPtr ctx, out long bytes_parked, out long blocks_parked);

        // ---- several processes, one per GPU (a .NET host normally uses stan_hip_init_multi instead)
        [DllImport(Lib)] internal static extern int stan_hip_comm_unique_id([Out] byte[] id);
        [DllImport(Lib)] internal static extern int stan_hip_comm_init(IntPtr ctx, int rank, int nranks, byte[] id);
        [DllImport(Lib)] internal static extern int stan_hip_comm_info(IntPtr ctx, out int rccl_version, out int comm_ranks, out int comm_rank, out int p2p);
        [DllImport(Lib)] internal static extern int stan_hip_comm_library(IntPtr ctx, [Out] byte[] path, long capacity, out int reused);

        // ---- assembly: ParallelAssembly_K (SolverFunctions.cs:117-180)
        [DllImport(Lib)] internal static extern int stan_hip_assemble_hex8(
            IntPtr ctx, long n_nodes, double[] xyz, int[] node_dof, long n_elem, int[] conn, int[] elem_mat,
            byte[] elem_type, int n_mat, double[] mat_E_nu, long n_dof, int[] ndof_reduction, out IntPtr outK);
        [DllImport(Lib)] internal static extern int stan_hip_assemble_hex8_dev(
            IntPtr ctx, long n_nodes, IntPtr d_xyz, IntPtr d_node_dof, long n_elem, IntPtr d_conn, IntPtr d_elem_mat,
            IntPtr d_elem_type, int n_mat, double[] mat_E_nu, long n_dof, IntPtr d_ndof_reduction, out IntPtr outK);
        [DllImport(Lib)] internal static extern void stan_hip_matrix_free(IntPtr K);

        // ---- solve: LinearSolver_CG (SolverFunctions.cs:270-330)
        [DllImport(Lib)] internal static extern int stan_hip_cg_solve(
            IntPtr ctx, IntPtr K, double[] F, double eps_f, int max_its, int precision_mode, [Out] double[] U,
            out int termination_type, out int iterations, out double rel_residual);
        [DllImport(Lib)] internal static extern int stan_hip_cg_solve_dev(
            IntPtr ctx, IntPtr K, IntPtr d_F, double eps_f, int max_its, int precision_mode, IntPtr d_U,
            out int termination_type, out int iterations, out double rel_residual);

        // ---- Element.Recovery_Stress + Update_StrainStress (Element.cs:211-246, 257-267)
        [DllImport(Lib)] internal static extern int stan_hip_recover_hex8(
            IntPtr ctx, long n_nodes, double[] xyz, double[] disp, long n_elem, int[] conn, int[] elem_mat,
            byte[] elem_type, int n_mat, double[] mat_E_nu, [Out] double[] strain, [Out] double[] stress);
        [DllImport(Lib)] internal static extern int stan_hip_recover_hex8_dev(
            IntPtr ctx, long n_nodes, IntPtr d_xyz, IntPtr d_disp, long n_elem, IntPtr d_conn, IntPtr d_elem_mat,
            IntPtr d_elem_type, int n_mat, double[] mat_E_nu, IntPtr d_strain, IntPtr d_stress);

        // ---- Element.Compute_NodalForces + R[DOF] += NodalForces (Element.cs:248-255, Solver.cs:187-196)
        [DllImport(Lib)] internal static extern int stan_hip_nodal_forces_hex8(
            IntPtr ctx, long n_nodes, double[] xyz, double[] disp, int[] node_dof, long n_elem, int[] conn,
            int[] elem_mat, byte[] elem_type, int n_mat, double[] mat_E_nu, long n_dof, [Out] double[] elem_forces,
            [Out] double[] R);

        // ---- introspection / parity helpers
        [DllImport(Lib)] internal static extern int stan_hip_matrix_info(IntPtr K, out StanMatrixInfo info);
        [DllImport(Lib)] internal static extern int stan_hip_recover_hex8_keep(
            IntPtr ctx, long n_nodes, double[] xyz, double[] disp, long n_elem, int[] conn, int[] elem_mat, byte[] elem_type,
            int n_mat, double[] mat_E_nu, out IntPtr results);
        [DllImport(Lib)] internal static extern int stan_hip_results_map(IntPtr results, long e0, long e1, out IntPtr strain, out IntPtr stress);
        [DllImport(Lib)] internal static extern void stan_hip_results_free(IntPtr results);
        [DllImport(Lib)] internal static extern int stan_hip_matrix_diagonal(IntPtr ctx, IntPtr K, [Out] double[] diag);
        [DllImport(Lib)] internal static extern int stan_hip_matrix_part_info(IntPtr K, int part, out StanMatrixInfo info);
        [DllImport(Lib)] internal static extern int stan_hip_ke_hex8(IntPtr ctx, double[] xyz8, double E, double nu, int type, [Out] double[] ke576);
        [DllImport(Lib)] internal static extern int stan_hip_ke_hex8_batch(IntPtr ctx, long n, double[] xyz8, double E, double nu, byte[] type, [Out] double[] ke);
        [DllImport(Lib)] internal static extern int stan_hip_matrix_to_csr(IntPtr ctx, IntPtr K, int upper_only, ref long nnz, [Out] long[] rowptr, [Out] int[] col, [Out] double[] val);
        [DllImport(Lib)] internal static extern int stan_hip_matrix_plan(
            IntPtr ctx, IntPtr K, [Out] long[] row_starts, out long n_halo, [Out] int[] halo_glob, out int n_nbr,
            [Out] int[] nbr, [Out] long[] send_off, [Out] int[] send_rows, [Out] long[] recv_off);
        [DllImport(Lib)] internal static extern int stan_hip_spmv_local(IntPtr ctx, IntPtr K, double[] x_local, [Out] double[] y_owned);
        [DllImport(Lib)] internal static extern int stan_hip_spmv(IntPtr ctx, IntPtr K, double[] x, [Out] double[] y);
        [DllImport(Lib)] internal static extern int stan_hip_spmv_bench(IntPtr ctx, IntPtr K, int precision_mode, int reps, out double avg_ms);
        [DllImport(Lib)] internal static extern int stan_hip_stream_bench(IntPtr ctx, IntPtr K, int reps, out double avg_ms, out long bytes);
        [DllImport(Lib)] internal static extern int stan_hip_set_profiling(IntPtr ctx, int enabled);
        [DllImport(Lib)] internal static extern int stan_hip_get_profile(IntPtr ctx, out StanProfile profile);
        [DllImport(Lib)] internal static extern int stan_hip_get_profile_rank(IntPtr ctx, int rank, out StanProfile profile);
        [DllImport(Lib)] internal static extern int stan_hip_device_info(IntPtr ctx, int rank, out int hip_ordinal, [Out] byte[] bus_id);
    }

    /// Owner of the context and of K (library-owned objects, explicit free).  One per solve in the reference's
    /// single-shot process; a host that solves repeatedly keeps the context and frees only K.
    public sealed class StanHipMatrix : IDisposable
    {
        internal IntPtr Ctx = IntPtr.Zero, K = IntPtr.Zero;
        // the flat arrays the assembly was called with: the stress recovery takes the same ones
        internal double[] Xyz, MatEnu;
        internal int[] NodeDof, Conn, ElemMat;
        internal byte[] ElemType;

        public void Dispose()
        {
            if (K != IntPtr.Zero) { StanHipNative.stan_hip_matrix_free(K); K = IntPtr.Zero; }
            if (Ctx != IntPtr.Zero) { StanHipNative.stan_hip_destroy(Ctx); Ctx = IntPtr.Zero; }
        }
    }
}
