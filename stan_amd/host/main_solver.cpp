// main_solver.cpp -- native console driver: `stan_solver <model.STdb>`.
// Reproduces Solver.Main / SolverLinearStatics / ExportOutput (Solver.cs:18-217, 454-462)
// on top of the two C-ABI libraries: read STdb -> materials -> AssignDOF -> BC tables ->
// assemble on the GPU -> CG on the GPU -> displacements -> stress recovery on the GPU ->
// overwrite the input file with the results (Result_StepNo = 1).
// Differences kept on purpose: no console-window sizing in the banner
// (SolverFunctions.cs:18-20 throws when stdout is redirected) and no 10 s sleep at exit
// (Solver.cs:67-68).  LinSolver "Cholesky"/"LU" (SolverFunctions.cs:332-516) are outside the GPU
// hot path: K is exported as the upper CRS the reference holds and solved on the CPU (direct.cpp).
// Extra switches (never stored in the STdb): --device N, --gpus N (devices 0..N-1) or --devices a,b,c
// (several GPUs from this ONE process: stan_hip_init_multi, rows of K sharded, RCCL inside the CG),
// --mixed, --fixed48, --no-merit-stop, --packed, --placement-tries N (default 16; 1 = plain allocation),
// --p2p (several GPUs: the CG exchanges peer to peer instead of over RCCL, STAN_OPT_COMM_P2P),
// --object-results (store the results in the Node / Element objects before the export, as the reference does,
// instead of encoding them from the flat arrays: same bytes), --json (one JSON line with sizes, iterations,
// device and host phase times and the SpMV's HBM rate).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/stan_hip.h"
#include "../../include/stan_host.h"
#include "model.h"
#include "solver_functions.h"

using namespace stan;
using clk = std::chrono::steady_clock;
static double secs(clk::time_point a) { return std::chrono::duration<double>(clk::now() - a).count(); }

// fn(i0, i1) over [0, n) on the host threads of libstan_host (STAN_HOST_THREADS)
template <typename F>
static void parallel_ranges(size_t n, F fn) {
    const size_t nt = (size_t)HostThreads();
    if (nt <= 1 || n < 4096) { fn((size_t)0, n); return; }
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; t++) th.emplace_back([=] { fn(n * t / nt, n * (t + 1) / nt); });
    for (std::thread &x : th) x.join();
}

static int fail(const char *what, const std::string &msg) {
    fprintf(stderr, "\n  ERROR in %s: %s\n", what, msg.c_str());
    return 1;
}

int main(int argc, char **argv) {
    std::string path;
    int device = 0, precision = STAN_PREC_FP64, placement_tries = 16;
    bool merit_stop = true, packed = false, json = false, object_results = false, p2p = false;
    std::vector<int> devices;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--gpus") && i + 1 < argc) {
            devices.clear();
            for (int d = 0, n = atoi(argv[++i]); d < n; d++) devices.push_back(d);
        } else if (!strcmp(argv[i], "--devices") && i + 1 < argc) {
            devices.clear();
            for (const char *p = argv[++i]; *p;) {
                devices.push_back(atoi(p));
                while (*p && *p != ',') p++;
                if (*p == ',') p++;
            }
        }
        else if (!strcmp(argv[i], "--placement-tries") && i + 1 < argc) placement_tries = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--mixed")) precision = STAN_PREC_MIXED;
        else if (!strcmp(argv[i], "--fixed48")) precision = STAN_PREC_FIXED48;
        else if (!strcmp(argv[i], "--no-merit-stop")) merit_stop = false;
        else if (!strcmp(argv[i], "--packed")) packed = true;
        else if (!strcmp(argv[i], "--json")) json = true;
        else if (!strcmp(argv[i], "--object-results")) object_results = true;
        else if (!strcmp(argv[i], "--p2p")) p2p = true;
        else path = argv[i];
    }
    if (path.empty()) {  // Path = path[0] -> IndexOutOfRangeException in the reference
        fprintf(stderr, "usage: stan_solver [--device N | --gpus N | --devices a,b,..] [--p2p] [--mixed|--fixed48] "
                        "[--no-merit-stop] [--packed] [--object-results] [--json] <model.STdb>\n");
        return 2;
    }
    SolverOptions opt;
    opt.device = device; opt.devices = devices; opt.precision = precision; opt.placement_tries = placement_tries; opt.merit_stop = merit_stop; opt.profile = json;
    opt.p2p = p2p;
    SolverFunctions Functions(opt);  // Solver.cs:16
    Functions.Welcome_Messsage();
    Functions.Prewarm();             // the device context comes up while the file is read

    // host phase times of this run (--json): the path around the GPU hot path is host work, and at scale most
    // of the wall clock
    double t_read = 0, t_dof = 0, t_bc = 0, t_disp = 0, t_recover = 0, t_store = 0, t_write = 0, t_hot = 0;
    const auto t_wall = clk::now();
    printf("   Reading input file: ");  // Solver.cs:23-41
    fflush(stdout);
    Database DB;
    std::string err;
    auto t0 = clk::now();
    if (!Functions.ProtoDeserialize(path, &DB, &err)) return fail("ProtoDeserialize", err);
    t_read = secs(t0);
    printf("     Done\n");

    printf("   DoF ordering: ");  // Solver.cs:44-47
    fflush(stdout);
    t0 = clk::now();
    if (int rc = DB.AssignDOF()) return fail("AssignDOF", "code " + std::to_string(rc) +
                                             " (disconnected mesh or unknown node ID)");
    t_dof = secs(t0);
    printf("           Done\n");
    fputs(DB.Database_Summary().c_str(), stdout);  // Solver.cs:50

    if (DB.AnalysisLib.Type == "Linear_Statics") try {  // Solver.cs:53-57, SolverLinearStatics :72-217
        const auto t_total = clk::now();
        const char *separator = "  ========================================================== ";
        printf("\n%s\n        LINEAR STATIC ANALYSIS \n%s\n", separator, separator);

        // Fix_DOF / nDOF_reduction / F (Solver.cs:104-152)
        std::vector<int32_t> nDOF_reduction;
        std::vector<double> F;
        int64_t n_fixed = 0;
        t0 = clk::now();
        if (BuildReductionAndLoads(DB, &nDOF_reduction, &n_fixed, &F, &err)) return fail("boundary conditions", err);
        t_bc = secs(t0);
        const int64_t n_nodes = (int64_t)DB.NodeLib.Count();

        t0 = clk::now();
        SparseMatrixHandle K;
        Functions.ParallelAssembly_K(DB, nDOF_reduction, 1, "Initial", &K);  // Solver.cs:157
        stan_matrix_info minfo;
        stan_hip_matrix_info(K.K, &minfo);

        std::vector<double> U((size_t)(DB.nDOF - n_fixed), 0.0);
        const std::string &ls = DB.AnalysisLib.LinSolver;
        if (ls == "CG") {  // Solver.cs:162
            U = Functions.LinearSolver_CG(K, F, DB.AnalysisLib);
            printf("   CG iterations: %d, scaled relative residual %.3e\n", Functions.last_iterations,
                   Functions.last_rel_residual);
        } else if (ls == "Cholesky") {  // Solver.cs:163: CPU fallback, outside the GPU hot path (direct.cpp)
            if (devices.size() > 1) return fail("solver selection", "the direct solvers run on one device's export of K: drop --gpus/--devices");
            U = Functions.LinearSolver_Cholesky(K, F);
        } else if (ls == "LU") {        // Solver.cs:164
            if (devices.size() > 1) return fail("solver selection", "the direct solvers run on one device's export of K: drop --gpus/--devices");
            U = Functions.LinearSolver_LU(K, F);
        }  // any other string: the reference leaves U = 0 (Solver.cs:160-164)
        t_hot = secs(t0);   // flatten + context + upload + assembly + solve + download

        // U = Include_BC_DOF(U, nDOF_reduction); node.dU_buffer[d] = U[DOF[d]] (Solver.cs:168-178)
        t0 = clk::now();
        const std::vector<double> Ufull = Functions.Include_BC_DOF(U, nDOF_reduction);
        std::vector<double> disp((size_t)n_nodes * 3);
        parallel_ranges(disp.size(), [&](size_t a, size_t b) {
            for (size_t k = a; k < b; k++) disp[k] = Ufull[(size_t)K.flat.node_dof[k]];
        });
        t_disp = secs(t0);

        printf("   Stress recovery: ");  // Solver.cs:183
        fflush(stdout);
        t0 = clk::now();
        std::vector<double> strain, stress;
        // the flat-array export takes the element results from the device chunk by chunk while it encodes (the
        // download overlaps the encoding and the file writes); --object-results needs them as whole host arrays
        if (object_results) Functions.Recovery_Stress(K, disp, &strain, &stress);
        else Functions.Recovery_Stress_Keep(K, disp);
        t_recover = secs(t0);
        printf("            Done\n");
        stan_profile pr{};
        if (json) stan_hip_get_profile(K.ctx, &pr);
        // Solver.cs:81-90 (initialise step 0/1), :203-210 (update), Main :56
        t0 = clk::now();
        // Solver.cs:81-90, 203-210: every node / element object is initialised and updated with the results, then
        // ExportOutput serialises the objects.  By default the writer takes the results from the flat arrays
        // instead (Database::ResultView: the same bytes without 13 million small heap objects);
        // --object-results walks the reference's object path (tests/test_gpu_parity.py compares the files).
        if (!object_results) {
            DB.results.disp = disp.data();
            stan_results *res = K.results;
            DB.results.fetch = [res](size_t e0, size_t e1, const double **sn, const double **ss) {
                return stan_hip_results_map(res, (int64_t)e0, (int64_t)e1, sn, ss) == STAN_OK;
            };
        }
        auto &nodes = DB.NodeLib.Items();
        if (object_results) parallel_ranges(nodes.size(), [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) {
                Node &n = nodes[i].second;
                n.Initialize_StepZero();
                n.Initialize_NewDisp(1);
                for (int d = 0; d < 3; d++) n.dU_buffer[d] = disp[3 * i + (size_t)d];
                n.Update_Displacement(1);
            }
        });
        auto &elems = DB.ElemLib.Items();
        if (object_results) parallel_ranges(elems.size(), [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) {
                Element &e = elems[i].second;
                e.Initialize_StepZero();
                e.Initialize_Increment(1);
                memcpy(e.Strain[1].M.data(), &strain[48 * i], 48 * sizeof(double));
                memcpy(e.Stress[1].M.data(), &stress[48 * i], 48 * sizeof(double));
            }
        });
        t_store = secs(t0);
        printf("\n%s\n  Total CPU time: %.2f s\n%s\n", separator, secs(t_total), separator);
        DB.AnalysisLib.Result_StepNo = 1;
        // ExportOutput: overwrite the input path (Solver.cs:454-462)
        t0 = clk::now();
        if (!Functions.ProtoSerialize(DB, path, packed, &err)) return fail("ExportOutput", err);
        t_write = secs(t0);
        if (json) {  // one machine-readable line per run (metrics/logging), after the export
            const double spmv_ms = pr.spmv_launches ? pr.spmv_ms_total / (double)pr.spmv_launches : 0;
            const double wall = secs(t_wall), dev = (pr.assemble_ms + pr.cg_ms) * 1e-3;
            printf("{\"n_gpus\": %d, \"n_dof\": %d, \"n_reduced\": %lld, \"blocks_3x3\": %lld, \"cg_iterations\": %d, "
                   "\"termination_type\": %d, \"rel_residual\": %.3e, \"t_assembly_s\": %.4f, \"t_cg_s\": %.4f, "
                   "\"spmv_ms\": %.4f, \"spmv_GBs\": %.1f, \"hbm_frac\": %.3f, "
                   "\"host_threads\": %d, \"t_wall_s\": %.3f, \"t_device_assembly_plus_cg_s\": %.4f, "
                   "\"phases_s\": {\"read_parse\": %.3f, \"assign_dof\": %.3f, \"bc_tables\": %.3f, "
                   "\"flatten_upload_assemble_solve\": %.3f, \"displacements\": %.3f, \"stress_recovery\": %.3f, "
                   "\"store_results\": %.3f, \"serialize_write\": %.3f}}\n",
                   devices.size() > 1 ? (int)devices.size() : 1, DB.nDOF, (long long)minfo.n_reduced, (long long)minfo.n_blocks, Functions.last_iterations,
                   Functions.last_termination_type, Functions.last_rel_residual, Functions.last_assembly_s,
                   Functions.last_cg_s, spmv_ms, spmv_ms > 0 ? pr.spmv_bytes / spmv_ms / 1e6 : 0.0,
                   spmv_ms > 0 ? pr.spmv_bytes / spmv_ms / 1e6 / 8000.0 : 0.0,
                   HostThreads(), wall, dev, t_read, t_dof, t_bc, t_hot, t_disp, t_recover, t_store, t_write);
        }
        return 0;
    } catch (const std::exception &e) {  // the C# lets these escape Main as unhandled exceptions
        return fail("SolverLinearStatics", e.what());
    } else if (DB.AnalysisLib.Type == "Nonlinear_Statics") {
        return fail("analysis type", "Nonlinear_Statics is unreachable from the GUI "
                                     "(MainWindow.xaml.cs:444) and outside the hot path");
    }

    // ExportOutput: overwrite the input path (Solver.cs:454-462) -- also when the analysis type is not one
    // this driver solves (Main writes the file back regardless)
    if (!Functions.ProtoSerialize(DB, path, packed, &err)) return fail("ExportOutput", err);
    return 0;
}
