// SolverFunctions.Hip.cs -- the managed side of the drop-in: the two methods SolverLinearStatics calls
// (Solver.cs:156 ParallelAssembly_K, :162 LinearSolver_CG) and the stress-recovery loop (Solver.cs:183-210), same
// names, same argument meaning, same console lines, on libstan_hip.so.  New file for src/STAN_Solver/; the
// reference's own SolverFunctions.cs is not edited (its managed versions stay: they serve "Cholesky" and "LU").
// Solver.cs changes in two places, shown at the end of this file and in INTEGRATION.md section 3.
using System;
using System.Collections.Generic;
using System.Diagnostics;
using System.Globalization;
using System.Runtime.InteropServices;
using STAN_Database;

namespace STAN_Solver
{
    public class SolverFunctionsHip
    {
        /// What the last LinearSolver_CG call took (alglib.lincgreport.iterationscount / r2): kept, never printed --
        /// the console lines are the reference's, literal by literal (tests/test_integration_shim.py).
        public int LastIterations;
        public double LastRelResidual;

        /// STAN_GPUS = n > 1 in the environment: one handle for n devices (stan_hip_init_multi); default: device 0.
        static int GpuCount()
        {
            int n;
            return int.TryParse(Environment.GetEnvironmentVariable("STAN_GPUS"), out n) && n > 1 ? n : 1;
        }

        /// Replaces SolverFunctions.ParallelAssembly_K (SolverFunctions.cs:117-180).  `inc` and `type` are accepted
        /// for signature parity; linear statics only ever passes (1, "Initial") (Solver.cs:156).
        public StanHipMatrix ParallelAssembly_K(Database DB, int[] nDOF_reduction, int inc, string type)
        {
            Stopwatch sw = new Stopwatch(); sw.Start();
            Console.Write("   K Matrix assembly: ");                                   // SolverFunctions.cs:127
            var h = new StanHipMatrix();
            // flatten the object graph in wire (Dictionary) order: that order drives AssignDOF and the element order
            var nodeIndex = new Dictionary<int, int>(DB.NodeLib.Count);                // Node ID -> position
            h.Xyz = new double[3 * DB.NodeLib.Count];
            h.NodeDof = new int[3 * DB.NodeLib.Count];
            int i = 0;
            foreach (Node n in DB.NodeLib.Values)
            {
                nodeIndex[n.ID] = i;
                h.Xyz[3 * i] = n.X; h.Xyz[3 * i + 1] = n.Y; h.Xyz[3 * i + 2] = n.Z;
                h.NodeDof[3 * i] = n.DOF[0]; h.NodeDof[3 * i + 1] = n.DOF[1]; h.NodeDof[3 * i + 2] = n.DOF[2];
                i++;
            }
            var matIndex = new Dictionary<int, int>();
            h.MatEnu = new double[2 * DB.MatLib.Count];
            i = 0;
            foreach (Material m in DB.MatLib.Values)
            {
                matIndex[m.ID] = i; h.MatEnu[2 * i] = m.E; h.MatEnu[2 * i + 1] = m.Poisson; i++;
            }
            h.Conn = new int[8 * DB.ElemLib.Count];
            h.ElemMat = new int[DB.ElemLib.Count];
            h.ElemType = new byte[DB.ElemLib.Count];
            i = 0;
            foreach (Element e in DB.ElemLib.Values)
            {
                if (e.Type != "HEX8_G1" && e.Type != "HEX8_G2")                       // Database.cs:44-48 admits CHEXA only
                    throw new InvalidOperationException("stan_hip: element " + e.ID + " has type " + e.Type);
                for (int a = 0; a < 8; a++) h.Conn[8 * i + a] = nodeIndex[e.NList[a]];
                h.ElemMat[i] = matIndex[e.MatID];                                      // MatLib[MatID], Element.cs:147
                h.ElemType[i] = e.Type == "HEX8_G1" ? StanHipNative.STAN_HEX8_G1 : StanHipNative.STAN_HEX8_G2;
                i++;
            }
            int gpus = GpuCount();
            Check(IntPtr.Zero, gpus > 1 ? StanHipNative.stan_hip_init_multi(gpus, null, out h.Ctx)
                                        : StanHipNative.stan_hip_init(0, out h.Ctx));
            Check(h.Ctx, StanHipNative.stan_hip_assemble_hex8(h.Ctx, DB.NodeLib.Count, h.Xyz, h.NodeDof,
                  DB.ElemLib.Count, h.Conn, h.ElemMat, h.ElemType, DB.MatLib.Count, h.MatEnu,
                  DB.nDOF, nDOF_reduction, out h.K));
            sw.Stop();
            Console.WriteLine("          Done in " + sw.Elapsed.TotalSeconds.ToString("F2", CultureInfo.InvariantCulture) + "s");
            return h;
        }

        /// Replaces SolverFunctions.LinearSolver_CG (SolverFunctions.cs:270-330): alglib.lincg's defaults, stopping
        /// rules and termination codes are the library's defaults; U is returned whatever the code (:329).
        public double[] LinearSolver_CG(StanHipMatrix K, double[] F, Analysis AnalysisLib)
        {
            Stopwatch sw = new Stopwatch(); sw.Start();
            Console.Write("   Solving linear system...   ");                           // SolverFunctions.cs:273
            double[] U = new double[F.Length];
            int type, its; double rel;
            Check(K.Ctx, StanHipNative.stan_hip_cg_solve(K.Ctx, K.K, F,
                  AnalysisLib.GetLinSolverTolerance(), AnalysisLib.GetLinSolverMaxIter(), StanHipNative.STAN_PREC_FP64,
                  U, out type, out its, out rel));
            Console.Write(type == 1 || type == 7 ? "  NORMAL " : "  ERROR ");          // :308-325
            Console.Write(" (type " + type + ")");                                     // :325, verbatim
            sw.Stop();
            Console.WriteLine(" in " + sw.Elapsed.TotalSeconds.ToString("F2", CultureInfo.InvariantCulture) + "s");
            LastIterations = its; LastRelResidual = rel;   // (alglib's report carries them too; the reference prints neither)
            return U;
        }

        /// Replaces the Parallel.ForEach of Solver.cs:184-197 (Element.Recovery_Stress; Compute_NodalForces only feeds
        /// R, which SolverLinearStatics discards at :199).  Call AFTER n.dU_buffer has been filled (Solver.cs:171-178)
        /// and BEFORE the Update_StrainStress loop (:206-209), which then copies dE/dS into Strain[inc]/Stress[inc]
        /// exactly as before.  Element.K_Initial's J[g]/BL[g] caches (Element.cs:127,143), which the managed
        /// Recovery_Stress reads, are not filled by the native assembly: this method must replace that loop, not
        /// run next to it.
        public void Recovery_Stress(Database DB, StanHipMatrix K)
        {
            Console.Write("   Stress recovery: ");                                     // Solver.cs:183
            double[] disp = new double[3 * DB.NodeLib.Count];
            int q = 0;
            foreach (Node n in DB.NodeLib.Values)
            {
                disp[q++] = n.dU_buffer[0]; disp[q++] = n.dU_buffer[1]; disp[q++] = n.dU_buffer[2];
            }
            double[] strain = new double[48 * DB.ElemLib.Count], stress = new double[48 * DB.ElemLib.Count];
            // a HEX8_G1 element returns STAN_E_UNSUPPORTED here, where the managed code throws (N has one row, Element.cs:242)
            Check(K.Ctx, StanHipNative.stan_hip_recover_hex8(K.Ctx, DB.NodeLib.Count, K.Xyz, disp,
                  DB.ElemLib.Count, K.Conn, K.ElemMat, K.ElemType, DB.MatLib.Count, K.MatEnu, strain, stress));
            // write-back: row a of element i's 8x6 block -> the 6x1 increments dE[a] / dS[a] that
            // Element.Update_StrainStress (Element.cs:257-267) copies into Strain[inc] / Stress[inc].
            // Initialize_Increment (Element.cs:92-107) has allocated them as zeros; the managed code adds into
            // them, so setting them is the same state.
            int i = 0;
            foreach (Element e in DB.ElemLib.Values)
            {
                for (int a = 0; a < 8; a++)
                    for (int c = 0; c < 6; c++)
                    {
                        e.dE[a].SetFast(c, 0, strain[48 * i + 6 * a + c]);
                        e.dS[a].SetFast(c, 0, stress[48 * i + 6 * a + c]);
                    }
                i++;
            }
            Console.WriteLine("            Done");                                     // Solver.cs:200
        }

        static void Check(IntPtr ctx, int rc)
        {
            if (rc == StanHipNative.STAN_OK) return;
            string msg = Marshal.PtrToStringAnsi(StanHipNative.stan_hip_last_error(ctx));
            // det J == 0: the managed code throws ArgumentException from MatrixST.Inverse (MatrixST.cs:315-318)
            if (rc == StanHipNative.STAN_E_DETJ) throw new ArgumentException("Inverse matrix error: " + msg, "Inverse");
            throw new InvalidOperationException("stan_hip error " + rc + ": " + msg);
        }
    }
}

/*  The change in Solver.cs (SolverLinearStatics), two places; `FunHip` is a `static SolverFunctionsHip` next to `Fun`:

    -- lines 156-164 (assembly + solve) become
        double[] U = new double[DB.nDOF - Fix_DOF.Count];
        StanHipMatrix Kh = null;
        if (DB.AnalysisLib.GetLinSolver() == "CG")
        {
            Kh = FunHip.ParallelAssembly_K(DB, nDOF_reduction, inc, "Initial");
            U = FunHip.LinearSolver_CG(Kh, F, DB.AnalysisLib);
        }
        else
        {
            alglib.sparsematrix K = Fun.ParallelAssembly_K(DB, nDOF_reduction, inc, "Initial");
            if (DB.AnalysisLib.GetLinSolver() == "Cholesky") U = Fun.LinearSolver_Cholesky(K, F);
            if (DB.AnalysisLib.GetLinSolver() == "LU")       U = Fun.LinearSolver_LU(K, F);
        }

    -- lines 183-200 (stress recovery) become
        if (Kh != null) { FunHip.Recovery_Stress(DB, Kh); Kh.Dispose(); }
        else { ... the existing Parallel.ForEach, unchanged ... }

    Everything else of SolverLinearStatics (Fix_DOF, nDOF_reduction, F, Include_BC_DOF, dU_buffer, Update_Displacement,
    Update_StrainStress, the timing lines) stays as it is.  */
