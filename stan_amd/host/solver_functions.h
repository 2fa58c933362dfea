// solver_functions.h -- C++ mirror of STAN_Solver.SolverFunctions for the linear-static path
// (SolverFunctions.cs): same method names, argument meaning, console lines and error
// behaviour, with the two hot calls going through the C-ABI of libstan_hip.so.
//   ParallelAssembly_K   SolverFunctions.cs:117-180  -> stan_hip_assemble_hex8
//   LinearSolver_CG      SolverFunctions.cs:270-330  -> stan_hip_cg_solve
//   Include_BC_DOF       SolverFunctions.cs:520-538
//   Exclude_BC_DOF       SolverFunctions.cs:540-555
//   Vector_Norm          SolverFunctions.cs:559-569
//   ProtoSerialize / ProtoDeserialize  :48-63      -> stdb.cpp
// Failures that make the C# throw (det J == 0 in MatrixST.Inverse, KeyNotFound on MatID ...)
// throw std::runtime_error here; a CG that does not converge is reported, not thrown, and U is
// returned regardless (SolverFunctions.cs:323-329).
#pragma once

#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/stan_hip.h"
#include "model.h"

namespace stan {

// alglib.sparsematrix in the reference: here an opaque device-resident K plus its context.
class SparseMatrixHandle {
  public:
    SparseMatrixHandle() {}
    ~SparseMatrixHandle();
    stan_results *results = nullptr;   // strain / stress kept on the device(s) until the export has encoded them
    SparseMatrixHandle(const SparseMatrixHandle &) = delete;
    SparseMatrixHandle &operator=(const SparseMatrixHandle &) = delete;
    stan_ctx *ctx = nullptr;
    stan_matrix *K = nullptr;
    FlatModel flat;  // kept for Recovery_Stress (the reference caches J/BL on the elements)
};

struct SolverOptions {  // extras of the native driver, never stored in the STdb
    int device = 0;
    std::vector<int> devices;   // --gpus N / --devices a,b,...: one process drives them all (stan_hip_init_multi)
    int precision = STAN_PREC_FP64;
    bool merit_stop = true;
    bool profile = false;
    bool p2p = false;           // STAN_OPT_COMM_P2P on a multi-device handle (throws where the devices cannot reach each other)
    int placement_tries = 16;   // STAN_OPT_PLACEMENT_TRIES: K's value stream is placed by search (only blocks >= 256 MB)
};

class SolverFunctions {
  public:
    explicit SolverFunctions(const SolverOptions &opt = SolverOptions()) : opt_(opt) {}
    ~SolverFunctions();

    // Round 5: the device context (HIP runtime start-up, streams, the communicator of a multi-device handle) comes up
    // on a thread of its own while the input file is read; ParallelAssembly_K waits for it.  Not calling it is fine.
    void Prewarm();

    void Welcome_Messsage() const;
    bool ProtoDeserialize(const std::string &path, Database *db, std::string *err) const { return ReadStdb(path, db, err); }
    bool ProtoSerialize(const Database &db, const std::string &path, bool packed, std::string *err) const { return WriteStdb(db, path, packed, err); }

    // K = ParallelAssembly_K(DB, nDOF_reduction, inc, "Initial")
    void ParallelAssembly_K(const Database &DB, const std::vector<int32_t> &nDOF_reduction, int inc,
                            const std::string &type, SparseMatrixHandle *K) const;
    // U = LinearSolver_CG(K, F, AnalysisLib)
    std::vector<double> LinearSolver_CG(SparseMatrixHandle &K, const std::vector<double> &F,
                                        const Analysis &AnalysisLib) const;
    // U = LinearSolver_Cholesky(K, F) / LinearSolver_LU(K, F) (SolverFunctions.cs:332-516): the
    // direct solvers are outside the GPU hot path -- K is exported as the reduced upper CRS alglib
    // would hold and factorised on the CPU (libstan_host.so, direct.cpp)
    std::vector<double> LinearSolver_Cholesky(SparseMatrixHandle &K, const std::vector<double> &F) const;
    std::vector<double> LinearSolver_LU(SparseMatrixHandle &K, const std::vector<double> &F) const;
    // Element.Recovery_Stress + Update_StrainStress for every element (Solver.cs:183-210)
    void Recovery_Stress(SparseMatrixHandle &K, const std::vector<double> &nodal_dU,
                         std::vector<double> *strain, std::vector<double> *stress) const;
    // the same with the results left on the device(s) (K.results): the export maps them chunk by chunk
    // (stan_hip_results_map through Database::ResultView::fetch) instead of holding 1.5 KB per element on the host
    void Recovery_Stress_Keep(SparseMatrixHandle &K, const std::vector<double> &nodal_dU) const;

    std::vector<double> Include_BC_DOF(const std::vector<double> &A, const std::vector<int32_t> &nDOF_reduction) const;
    std::vector<double> Exclude_BC_DOF(const std::vector<double> &A, const std::vector<int32_t> &nDOF_reduction) const;
    double Vector_Norm(const std::vector<double> &v) const;

    // report of the last LinearSolver_CG
    int last_termination_type = 0, last_iterations = 0;
    double last_rel_residual = 0, last_assembly_s = 0, last_cg_s = 0;

  private:
    SolverOptions opt_;
    struct Warm;            // the context being created ahead of its use
    Warm *warm_ = nullptr;
};

}  // namespace stan
