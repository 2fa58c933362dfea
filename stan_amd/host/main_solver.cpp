// main_solver.cpp -- native console driver: `stan_solver <model.STdb>`.
// Reproduces Solver.Main / SolverLinearStatics / ExportOutput (Solver.cs:18-217, 454-462)
// on top of the two C-ABI libraries: read STdb -> materials -> AssignDOF -> BC tables ->
// assemble on the GPU -> CG on the GPU -> displacements -> stress recovery on the GPU ->
// overwrite the input file with the results (Result_StepNo = 1).
// Differences kept on purpose: no console-window sizing in the banner
// (SolverFunctions.cs:18-20 throws when stdout is redirected) and no 10 s sleep at exit
// (Solver.cs:67-68).  LinSolver "Cholesky"/"LU" (SolverFunctions.cs:332-516) are outside the
// hot path: the driver reports them as unsupported instead of silently using CG.
// Extra switches (never stored in the STdb): --device N, --mixed, --no-merit-stop, --packed,
// --json (one JSON line with sizes, iterations, phase times and the SpMV's HBM rate).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/stan_hip.h"
#include "../../include/stan_host.h"
#include "model.h"

using namespace stan;
using clk = std::chrono::steady_clock;
static double secs(clk::time_point a) { return std::chrono::duration<double>(clk::now() - a).count(); }

static void Welcome_Messsage() {  // SolverFunctions.cs:23-44
    puts("");
    puts("  ========================================================== ");
    puts("  ********************************************************** ");
    puts("                  STAN - STructural ANalyser                 ");
    puts("  ********************************************************** ");
    puts("      Solver: Linear, Statics  (MI355X native hot path)      ");
    puts("  ========================================================== ");
    puts("                                                             ");
}

static int fail(const char *what, const std::string &msg) {
    fprintf(stderr, "\n  ERROR in %s: %s\n", what, msg.c_str());
    return 1;
}

int main(int argc, char **argv) {
    std::string path;
    int device = 0, precision = STAN_PREC_FP64;
    bool merit_stop = true, packed = false, json = false;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--mixed")) precision = STAN_PREC_MIXED;
        else if (!strcmp(argv[i], "--no-merit-stop")) merit_stop = false;
        else if (!strcmp(argv[i], "--packed")) packed = true;
        else if (!strcmp(argv[i], "--json")) json = true;
        else path = argv[i];
    }
    if (path.empty()) {  // Path = path[0] -> IndexOutOfRangeException in the reference
        fprintf(stderr, "usage: stan_solver [--device N] [--mixed] [--no-merit-stop] <model.STdb>\n");
        return 2;
    }
    Welcome_Messsage();

    printf("   Reading input file: ");  // Solver.cs:23-41
    Database DB;
    std::string err;
    if (!ReadStdb(path, &DB, &err)) return fail("ProtoDeserialize", err);
    printf("     Done\n");

    printf("   DoF ordering: ");  // Solver.cs:44-47
    if (int rc = DB.AssignDOF()) return fail("AssignDOF", "code " + std::to_string(rc) +
                                             " (disconnected mesh or unknown node ID)");
    printf("           Done\n");
    fputs(DB.Database_Summary().c_str(), stdout);  // Solver.cs:50

    if (DB.AnalysisLib.Type == "Linear_Statics") {  // Solver.cs:53-57
        const auto t_total = clk::now();
        const char *separator = "  ========================================================== ";
        printf("\n%s\n        LINEAR STATIC ANALYSIS \n%s\n", separator, separator);

        FlatModel fm;
        if (Flatten(DB, &fm, &err)) return fail("model", err);
        std::vector<int32_t> red;
        std::vector<double> F;
        int64_t n_fixed = 0;
        if (BuildReductionAndLoads(DB, &red, &n_fixed, &F, &err)) return fail("boundary conditions", err);
        const int64_t n_nodes = (int64_t)DB.NodeLib.Count(), n_elem = (int64_t)DB.ElemLib.Count();

        stan_ctx *ctx = nullptr;
        if (stan_hip_init(device, &ctx)) return fail("stan_hip_init", stan_hip_last_error(nullptr));
        stan_hip_set_option(ctx, STAN_OPT_CG_MERIT_STOP, merit_stop ? 1 : 0);
        if (json) stan_hip_set_profiling(ctx, 1);
        double t_asm = 0, t_cg = 0;
        int32_t cg_type = 0, cg_its = 0;
        double cg_rel = 0;

        printf("   K Matrix assembly: ");  // SolverFunctions.cs:127
        fflush(stdout);
        auto t0 = clk::now();
        stan_matrix *K = nullptr;
        if (stan_hip_assemble_hex8(ctx, n_nodes, fm.xyz.data(), fm.node_dof.data(), n_elem,
                                   fm.conn.data(), fm.elem_mat.data(), fm.elem_type.data(),
                                   (int32_t)(fm.mat_E_nu.size() / 2), fm.mat_E_nu.data(), DB.nDOF,
                                   red.data(), &K))
            return fail("ParallelAssembly_K", stan_hip_last_error(ctx));
        t_asm = secs(t0);
        printf("          Done in %.2fs\n", t_asm);
        stan_matrix_info minfo;
        stan_hip_matrix_info(K, &minfo);

        std::vector<double> U((size_t)(DB.nDOF - n_fixed), 0.0);
        const std::string &ls = DB.AnalysisLib.LinSolver;
        if (ls == "CG") {  // Solver.cs:162
            printf("   Solving linear system...   ");  // SolverFunctions.cs:273
            fflush(stdout);
            t0 = clk::now();
            int32_t type = 0, its = 0;
            double rel = 0;
            if (stan_hip_cg_solve(ctx, K, F.data(), DB.AnalysisLib.LinSolverTolerance,
                                  DB.AnalysisLib.LinSolverIterMax, precision, U.data(), &type, &its, &rel))
                return fail("LinearSolver_CG", stan_hip_last_error(ctx));
            printf(type == 1 || type == 7 ? "  NORMAL " : "  ERROR ");  // SolverFunctions.cs:323-327
            t_cg = secs(t0);
            cg_type = type; cg_its = its; cg_rel = rel;
            printf(" (type %d) in %.2fs\n", type, t_cg);
            printf("   CG iterations: %d, scaled relative residual %.3e\n", its, rel);
        } else if (ls == "Cholesky" || ls == "LU") {
            return fail("solver selection", "LinSolver '" + ls + "' is a direct solver outside the GPU hot path");
        }  // any other string: the reference leaves U = 0 (Solver.cs:160-164)
        stan_hip_matrix_free(K);

        // Include_BC_DOF + write-back (SolverFunctions.cs:520-538, Solver.cs:168-178)
        std::vector<double> disp((size_t)n_nodes * 3);
        stan_host_nodal_displacements(n_nodes, fm.node_dof.data(), red.data(), U.data(), disp.data());

        printf("   Stress recovery: ");  // Solver.cs:183
        fflush(stdout);
        std::vector<double> strain((size_t)n_elem * 48), stress((size_t)n_elem * 48);
        if (stan_hip_recover_hex8(ctx, n_nodes, fm.xyz.data(), disp.data(), n_elem, fm.conn.data(),
                                  fm.elem_mat.data(), fm.elem_type.data(),
                                  (int32_t)(fm.mat_E_nu.size() / 2), fm.mat_E_nu.data(),
                                  strain.data(), stress.data()))
            return fail("Recovery_Stress", stan_hip_last_error(ctx));  // G1: the reference throws here too
        printf("            Done\n");
        if (json) {  // one machine-readable line per run (metrics/logging)
            stan_profile pr;
            stan_hip_get_profile(ctx, &pr);
            const double spmv_ms = pr.spmv_launches ? pr.spmv_ms_total / (double)pr.spmv_launches : 0;
            printf("{\"n_dof\": %d, \"n_reduced\": %lld, \"blocks_3x3\": %lld, \"cg_iterations\": %d, "
                   "\"termination_type\": %d, \"rel_residual\": %.3e, \"t_assembly_s\": %.4f, \"t_cg_s\": %.4f, "
                   "\"spmv_ms\": %.4f, \"spmv_GBs\": %.1f, \"hbm_frac\": %.3f}\n",
                   DB.nDOF, (long long)minfo.n_reduced, (long long)minfo.n_blocks, cg_its, cg_type, cg_rel, t_asm,
                   t_cg, spmv_ms, spmv_ms > 0 ? pr.spmv_bytes / spmv_ms / 1e6 : 0.0,
                   spmv_ms > 0 ? pr.spmv_bytes / spmv_ms / 1e6 / 8000.0 : 0.0);
        }
        stan_hip_destroy(ctx);

        // Solver.cs:81-90 (initialise step 0/1), :203-210 (update), Main :56
        size_t i = 0;
        for (auto &kv : DB.NodeLib.Items()) {
            Node &n = kv.second;
            n.Initialize_StepZero();
            n.Initialize_NewDisp(1);
            for (int d = 0; d < 3; d++) n.dU_buffer[d] = disp[3 * i + (size_t)d];
            n.Update_Displacement(1);
            i++;
        }
        i = 0;
        for (auto &kv : DB.ElemLib.Items()) {
            Element &e = kv.second;
            e.Initialize_StepZero();
            e.Initialize_Increment(1);
            memcpy(e.Strain[1].M.data(), &strain[48 * i], 48 * sizeof(double));
            memcpy(e.Stress[1].M.data(), &stress[48 * i], 48 * sizeof(double));
            i++;
        }
        printf("\n%s\n  Total CPU time: %.2f s\n%s\n", separator, secs(t_total), separator);
        DB.AnalysisLib.Result_StepNo = 1;
    } else if (DB.AnalysisLib.Type == "Nonlinear_Statics") {
        return fail("analysis type", "Nonlinear_Statics is unreachable from the GUI "
                                     "(MainWindow.xaml.cs:444) and outside the hot path");
    }

    // ExportOutput: overwrite the input path (Solver.cs:454-462)
    if (!WriteStdb(DB, path, packed, &err)) return fail("ExportOutput", err);
    return 0;
}
