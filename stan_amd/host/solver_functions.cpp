// solver_functions.cpp -- see solver_functions.h.
#include "solver_functions.h"

#include "../../include/stan_host.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <thread>

namespace stan {

using clk = std::chrono::steady_clock;
static double secs(clk::time_point a) { return std::chrono::duration<double>(clk::now() - a).count(); }

struct SolverFunctions::Warm {
    std::thread th;
    stan_ctx *ctx = nullptr;
    int rc = 0;
    std::string err;
};
SolverFunctions::~SolverFunctions() {
    if (!warm_) return;
    if (warm_->th.joinable()) warm_->th.join();
    if (warm_->ctx) stan_hip_destroy(warm_->ctx);   // never picked up (the run ended before the assembly)
    delete warm_;
}
void SolverFunctions::Prewarm() {
    if (warm_) return;
    warm_ = new Warm();
    Warm *w = warm_;
    const SolverOptions opt = opt_;
    w->th = std::thread([w, opt] {
        w->rc = opt.devices.size() > 1 ? stan_hip_init_multi((int)opt.devices.size(), opt.devices.data(), &w->ctx)
                                       : stan_hip_init(opt.devices.empty() ? opt.device : opt.devices[0], &w->ctx);
        if (w->rc) w->err = stan_hip_last_error(nullptr);   // (thread-local text: taken on this thread)
    });
}

SparseMatrixHandle::~SparseMatrixHandle() {
    if (results) stan_hip_results_free(results);
    if (K) stan_hip_matrix_free(K);
    if (ctx) stan_hip_destroy(ctx);
}

void SolverFunctions::Welcome_Messsage() const {  // SolverFunctions.cs:23-44 (no window sizing, :18-20)
    puts("");
    puts("  ========================================================== ");
    puts("  ********************************************************** ");
    puts("                  STAN - STructural ANalyser                 ");
    puts("  ********************************************************** ");
    puts("      Solver: Linear, Statics  (MI355X native hot path)      ");
    puts("  ========================================================== ");
    puts("                                                             ");
}

void SolverFunctions::ParallelAssembly_K(const Database &DB, const std::vector<int32_t> &red, int inc,
                                         const std::string &type, SparseMatrixHandle *K) const {
    if (type != "Initial" || inc != 1)  // "Tangent" belongs to the unfinished nonlinear driver
        throw std::runtime_error("ParallelAssembly_K: only type \"Initial\" at increment 1 (linear statics)");
    const auto t0 = clk::now();
    printf("   K Matrix assembly: ");  // SolverFunctions.cs:127
    fflush(stdout);
    std::string err;
    if (Flatten(DB, &K->flat, &err)) throw std::runtime_error(err);
    // several GPUs: still ONE process and the same calls -- the handle fans them out (stan_hip.h)
    if (warm_) {   // the context was started while the file was read (Prewarm)
        if (warm_->th.joinable()) warm_->th.join();
        if (warm_->rc) throw std::runtime_error(warm_->err);
        K->ctx = warm_->ctx;
        warm_->ctx = nullptr;
    } else {
        const int rc_init = opt_.devices.size() > 1
                                ? stan_hip_init_multi((int)opt_.devices.size(), opt_.devices.data(), &K->ctx)
                                : stan_hip_init(opt_.devices.empty() ? opt_.device : opt_.devices[0], &K->ctx);
        if (rc_init) throw std::runtime_error(stan_hip_last_error(nullptr));
    }
    stan_hip_set_option(K->ctx, STAN_OPT_CG_MERIT_STOP, opt_.merit_stop ? 1 : 0);
    stan_hip_set_option(K->ctx, STAN_OPT_POOL_MAX_BYTES, -1);   // this process owns its device(s)
    stan_hip_set_option(K->ctx, STAN_OPT_PLACEMENT_TRIES, opt_.placement_tries < 1 ? 1 : opt_.placement_tries);
    if (opt_.p2p && stan_hip_set_option(K->ctx, STAN_OPT_COMM_P2P, 1)) throw std::runtime_error(stan_hip_last_error(K->ctx));
    if (opt_.profile) stan_hip_set_profiling(K->ctx, 1);
    const FlatModel &f = K->flat;
    const int rc = stan_hip_assemble_hex8(K->ctx, (int64_t)DB.NodeLib.Count(), f.xyz.data(), f.node_dof.data(),
                                          (int64_t)DB.ElemLib.Count(), f.conn.data(), f.elem_mat.data(),
                                          f.elem_type.data(), (int32_t)(f.mat_E_nu.size() / 2),
                                          f.mat_E_nu.data(), DB.nDOF, red.data(), &K->K);
    if (rc) throw std::runtime_error(stan_hip_last_error(K->ctx));  // det J == 0: MatrixST.cs:315-318 throws
    const_cast<SolverFunctions *>(this)->last_assembly_s = secs(t0);
    printf("          Done in %.2fs\n", last_assembly_s);  // SolverFunctions.cs:177
}

std::vector<double> SolverFunctions::LinearSolver_CG(SparseMatrixHandle &K, const std::vector<double> &F,
                                                     const Analysis &A) const {
    const auto t0 = clk::now();
    printf("   Solving linear system...   ");  // SolverFunctions.cs:273
    fflush(stdout);
    std::vector<double> U(F.size(), 0.0);
    int32_t type = 0, its = 0;
    double rel = 0;
    if (stan_hip_cg_solve(K.ctx, K.K, F.data(), A.LinSolverTolerance, A.LinSolverIterMax, opt_.precision,
                          U.data(), &type, &its, &rel))
        throw std::runtime_error(stan_hip_last_error(K.ctx));
    SolverFunctions *self = const_cast<SolverFunctions *>(this);
    self->last_termination_type = type; self->last_iterations = its; self->last_rel_residual = rel;
    self->last_cg_s = secs(t0);
    printf(type == 1 || type == 7 ? "  NORMAL " : "  ERROR ");  // SolverFunctions.cs:323-324
    printf(" (type %d) in %.2fs\n", type, last_cg_s);          // :325-327
    return U;                                                    // :329, returned regardless
}

// alglib.sparsematrix as the reference holds it after ParallelAssembly_K: reduced upper CRS
static void ExportUpperCrs(SparseMatrixHandle &K, std::vector<int64_t> *rowptr, std::vector<int32_t> *col,
                           std::vector<double> *val) {
    int64_t nnz = 0;
    if (stan_hip_matrix_to_csr(K.ctx, K.K, 1, &nnz, nullptr, nullptr, nullptr))
        throw std::runtime_error(stan_hip_last_error(K.ctx));
    stan_matrix_info info;
    stan_hip_matrix_info(K.K, &info);
    rowptr->assign((size_t)info.n_reduced + 1, 0);
    col->assign((size_t)nnz, 0);
    val->assign((size_t)nnz, 0.0);
    if (stan_hip_matrix_to_csr(K.ctx, K.K, 1, &nnz, rowptr->data(), col->data(), val->data()))
        throw std::runtime_error(stan_hip_last_error(K.ctx));
}

std::vector<double> SolverFunctions::LinearSolver_Cholesky(SparseMatrixHandle &K, const std::vector<double> &F) const {
    const auto t0 = clk::now();
    puts("   Linear system K*U=F:");                       // SolverFunctions.cs:385
    std::vector<int64_t> rp; std::vector<int32_t> ci; std::vector<double> cv;
    ExportUpperCrs(K, &rp, &ci, &cv);                      // sparseconverttosks works on the same triangle
    printf("    - Cholesky decomposition:");               // :388
    fflush(stdout);
    std::vector<double> U(F.size(), 0.0);
    int32_t type = 0;
    int64_t profile = 0;
    const int rc = stan_host_cholesky_skyline_solve((int64_t)F.size(), rp.data(), ci.data(), cv.data(), F.data(),
                                                    U.data(), &type, &profile);
    if (rc == STAN_HOST_E_MEMORY)
        throw std::runtime_error("LinearSolver_Cholesky: the skyline profile (" + std::to_string(profile) +
                                 " entries) does not fit in memory; use LinSolver = CG");
    if (rc) throw std::runtime_error("LinearSolver_Cholesky: bad matrix (code " + std::to_string(rc) + ")");
    puts(type > 0 ? "   Done" : "   ERROR");                 // :390-397
    printf("    - Solving:");                               // :426
    printf(type > 0 ? "                  NORMAL termination" : "                  ERROR termination");
    printf(" (type %d)\n", type);                           // :430-438
    SolverFunctions *self = const_cast<SolverFunctions *>(this);
    self->last_termination_type = type; self->last_iterations = 0; self->last_rel_residual = 0;
    self->last_cg_s = secs(t0);
    printf("    Total time to solve K*U=F:  %.2fs\n", last_cg_s);   // :441
    return U;
}

std::vector<double> SolverFunctions::LinearSolver_LU(SparseMatrixHandle &K, const std::vector<double> &F) const {
    const auto t0 = clk::now();
    printf("   Solving linear system...   ");               // SolverFunctions.cs:449
    fflush(stdout);
    std::vector<int64_t> rp; std::vector<int32_t> ci; std::vector<double> cv;
    ExportUpperCrs(K, &rp, &ci, &cv);
    std::vector<double> U(F.size(), 0.0);
    int32_t type = 0;
    if (int rc = stan_host_lu_upper_solve((int64_t)F.size(), rp.data(), ci.data(), cv.data(), F.data(), U.data(), &type))
        throw std::runtime_error("LinearSolver_LU: bad matrix (code " + std::to_string(rc) + ")");
    SolverFunctions *self = const_cast<SolverFunctions *>(this);
    self->last_termination_type = type; self->last_iterations = 0; self->last_rel_residual = 0;
    self->last_cg_s = secs(t0);
    printf(type > 0 ? "NORMAL TERMINATION" : "ERROR TERMINATION");   // :506-507
    printf(" (type %d) in %.2fs\n", type, last_cg_s);               // :508-510
    return U;
}

void SolverFunctions::Recovery_Stress(SparseMatrixHandle &K, const std::vector<double> &dU,
                                      std::vector<double> *strain, std::vector<double> *stress) const {
    const FlatModel &f = K.flat;
    const int64_t n_nodes = (int64_t)(f.xyz.size() / 3), n_elem = (int64_t)f.elem_mat.size();
    strain->assign((size_t)n_elem * 48, 0.0);
    stress->assign((size_t)n_elem * 48, 0.0);
    if (stan_hip_recover_hex8(K.ctx, n_nodes, f.xyz.data(), dU.data(), n_elem, f.conn.data(), f.elem_mat.data(),
                              f.elem_type.data(), (int32_t)(f.mat_E_nu.size() / 2), f.mat_E_nu.data(),
                              strain->data(), stress->data()))
        throw std::runtime_error(stan_hip_last_error(K.ctx));  // HEX8_G1: the reference throws too (Element.cs:242)
}

void SolverFunctions::Recovery_Stress_Keep(SparseMatrixHandle &K, const std::vector<double> &dU) const {
    const FlatModel &f = K.flat;
    const int64_t n_nodes = (int64_t)(f.xyz.size() / 3), n_elem = (int64_t)f.elem_mat.size();
    if (K.results) { stan_hip_results_free(K.results); K.results = nullptr; }
    if (stan_hip_recover_hex8_keep(K.ctx, n_nodes, f.xyz.data(), dU.data(), n_elem, f.conn.data(), f.elem_mat.data(),
                                   f.elem_type.data(), (int32_t)(f.mat_E_nu.size() / 2), f.mat_E_nu.data(), &K.results))
        throw std::runtime_error(stan_hip_last_error(K.ctx));  // HEX8_G1: the reference throws too (Element.cs:242)
}

std::vector<double> SolverFunctions::Include_BC_DOF(const std::vector<double> &A, const std::vector<int32_t> &red) const {
    std::vector<double> full(red.size(), 0.0);  // SolverFunctions.cs:520-538
    for (size_t i = 0; i < red.size(); i++) full[i] = red[i] == -1 ? 0.0 : A[i - (size_t)red[i]];
    return full;
}
std::vector<double> SolverFunctions::Exclude_BC_DOF(const std::vector<double> &A, const std::vector<int32_t> &red) const {
    size_t nfix = 0;  // SolverFunctions.cs:540-555
    for (int32_t r : red) nfix += r == -1;
    std::vector<double> out(red.size() - nfix, 0.0);
    for (size_t i = 0; i < red.size(); i++)
        if (red[i] != -1) out[i - (size_t)red[i]] = A[i];
    return out;
}
double SolverFunctions::Vector_Norm(const std::vector<double> &v) const {
    double n = 0;  // SolverFunctions.cs:559-569
    for (double x : v) n += std::pow(x, 2);
    return std::sqrt(n);
}

}  // namespace stan
