// capi_db.cpp -- C-ABI over the STAN_Database mirror (include/stan_host.h, stan_db part).
#include <cstring>

#include "../../include/stan_host.h"
#include "model.h"

struct stan_db {
    stan::Database db;
    std::string err;
};

using namespace stan;

extern "C" {

int stan_host_db_new(stan_db **out) {
    if (!out) return STAN_HOST_E_ARG;
    *out = new stan_db();
    return STAN_HOST_OK;
}
void stan_host_db_free(stan_db *d) { delete d; }
const char *stan_host_db_last_error(stan_db *d) { return d ? d->err.c_str() : ""; }

int stan_host_db_read_stdb(stan_db *d, const char *path) {
    if (!d || !path) return STAN_HOST_E_ARG;
    if (!ReadStdb(path, &d->db, &d->err))
        return d->err.rfind("cannot", 0) == 0 ? STAN_HOST_E_IO : STAN_HOST_E_FORMAT;
    return STAN_HOST_OK;
}
int stan_host_db_parse_stdb(stan_db *d, const uint8_t *data, int64_t size) {
    if (!d || (!data && size) || size < 0) return STAN_HOST_E_ARG;
    return ParseStdb(data, (size_t)size, &d->db, &d->err) ? STAN_HOST_OK : STAN_HOST_E_FORMAT;
}
int stan_host_db_write_stdb(stan_db *d, const char *path, int32_t packed) {
    if (!d || !path) return STAN_HOST_E_ARG;
    return WriteStdb(d->db, path, packed != 0, &d->err) ? STAN_HOST_OK : STAN_HOST_E_IO;
}
int stan_host_db_write_stdb_with_results(stan_db *d, const char *path, int32_t packed, const double *disp,
                                         const double *strain, const double *stress) {
    if (!d || !path || !disp || !strain || !stress) return STAN_HOST_E_ARG;
    d->db.results.disp = disp; d->db.results.strain = strain; d->db.results.stress = stress;
    const int prev = d->db.AnalysisLib.Result_StepNo;
    d->db.AnalysisLib.Result_StepNo = 1;   // Solver.cs:56
    const bool ok = WriteStdb(d->db, path, packed != 0, &d->err);
    d->db.results = Database::ResultView();
    d->db.AnalysisLib.Result_StepNo = prev;
    return ok ? STAN_HOST_OK : STAN_HOST_E_IO;
}
int stan_host_db_serialize(stan_db *d, int32_t packed, uint8_t *buf, int64_t cap, int64_t *size) {
    if (!d || !size) return STAN_HOST_E_ARG;
    std::string s;
    SerializeStdb(d->db, packed != 0, &s);
    *size = (int64_t)s.size();
    if (buf) {
        if (cap < (int64_t)s.size()) return STAN_HOST_E_ARG;
        memcpy(buf, s.data(), s.size());
    }
    return STAN_HOST_OK;
}

int stan_host_db_read_bdf(stan_db *d, const char *path, int64_t *n_import_errors) {
    if (!d || !path) return STAN_HOST_E_ARG;
    if (!d->db.ReadNastranMesh(path, &d->err)) return STAN_HOST_E_IO;
    d->db.Set_nDOF();
    if (n_import_errors) *n_import_errors = (int64_t)d->db.Import_Error.size();
    return STAN_HOST_OK;
}

int stan_host_db_set_mesh(stan_db *d, int64_t n_nodes, const int32_t *node_ids, const double *xyz,
                          int64_t n_elem, const int32_t *elem_ids, const int32_t *elem_pids,
                          const int32_t *nlist8, const char *hex_type) {
    if (!d || n_nodes < 0 || n_elem < 0 || (n_nodes && (!node_ids || !xyz)) ||
        (n_elem && (!elem_ids || !nlist8)))
        return STAN_HOST_E_ARG;
    d->db.NodeLib.Clear();
    d->db.ElemLib.Clear();
    d->db.conn_index.clear();
    for (int64_t i = 0; i < n_nodes; i++) {
        Node n;
        n.ID = node_ids[i];
        n.X = xyz[3 * i]; n.Y = xyz[3 * i + 1]; n.Z = xyz[3 * i + 2];
        n.DOF = {0, 0, 0};
        n.DispX = {0.0}; n.DispY = {0.0}; n.DispZ = {0.0};
        if (!d->db.NodeLib.Add(n.ID, n)) { d->err = "duplicate node ID " + std::to_string(n.ID); return STAN_HOST_E_ARG; }
    }
    for (int64_t e = 0; e < n_elem; e++) {
        Element el;
        el.ID = elem_ids[e];
        el.PID = elem_pids ? elem_pids[e] : 1;
        el.Type = hex_type ? hex_type : "HEX8_G2";
        el.has_type = true;
        el.NList.assign(nlist8 + 8 * e, nlist8 + 8 * e + 8);
        if (!d->db.ElemLib.Add(el.ID, el)) { d->err = "duplicate element ID " + std::to_string(el.ID); return STAN_HOST_E_ARG; }
    }
    d->db.Set_nDOF();
    return STAN_HOST_OK;
}

int stan_host_db_add_material(stan_db *d, int32_t id, const char *name, double E, double nu) {
    if (!d) return STAN_HOST_E_ARG;
    Material m = Material::Create(id);
    if (name) { m.Name = name; m.has_name = true; }
    m.E = E; m.Poisson = nu;  // SetElastic, Material.cs:31-35
    if (!d->db.MatLib.Add(id, m)) { d->err = "duplicate material ID"; return STAN_HOST_E_ARG; }
    return STAN_HOST_OK;
}

int stan_host_db_assign_part(stan_db *d, int32_t pid, int32_t mat_id, const char *hex_type) {
    if (!d) return STAN_HOST_E_ARG;
    for (auto &kv : d->db.ElemLib.Items())
        if (kv.second.PID == pid) {
            kv.second.MatID = mat_id;                                          // Part.cs:767-774
            if (hex_type) { kv.second.Type = hex_type; kv.second.has_type = true; }  // Part.cs:658-673
        }
    PartInfo pi;
    pi.ColorID = pid % 9; pi.MatID = mat_id; pi.Name = "Part ID " + std::to_string(pid);
    pi.HEX_Type = hex_type ? hex_type : "HEX8_G2";
    pi.PENTA_Type = "PENTA6_G2"; pi.TET_Type = "TET4_G2";
    if (PartInfo *p = d->db.Info.InfoPart.Find(pid)) *p = pi;
    else d->db.Info.InfoPart.Add(pid, pi);
    d->db.Info.has_parts = true;
    d->db.has_info = true;
    return STAN_HOST_OK;
}

int stan_host_db_add_bc(stan_db *d, int32_t id, const char *name, const char *type, int64_t n,
                        const int32_t *node_ids, const double *vals) {
    if (!d || !type || n < 0 || (n && (!node_ids || !vals))) return STAN_HOST_E_ARG;
    BoundaryCondition bc;
    bc.Name = name ? name : ""; bc.has_name = true;
    bc.Type = type; bc.has_type = true;
    bc.ID = id; bc.ColorID = id % 9;
    for (int64_t i = 0; i < n; i++) {
        if (!d->db.NodeLib.ContainsKey(node_ids[i])) continue;  // BoundaryCondition.cs:89
        MatrixST v(3, 1);
        v.M[0] = vals[3 * i]; v.M[1] = vals[3 * i + 1]; v.M[2] = vals[3 * i + 2];
        if (!bc.NodalValues.Add(node_ids[i], v)) { d->err = "BC lists node " + std::to_string(node_ids[i]) + " twice (Dictionary.Add throws)"; return STAN_HOST_E_ARG; }
    }
    if (!d->db.BCLib.Add(id, bc)) { d->err = "duplicate BC ID"; return STAN_HOST_E_ARG; }
    return STAN_HOST_OK;
}

int stan_host_db_set_analysis(stan_db *d, const char *type, const char *lin_solver, double tol,
                              int32_t max_iter, int32_t inc_numb) {
    if (!d) return STAN_HOST_E_ARG;
    Analysis &a = d->db.AnalysisLib;
    if (type) a.Type = type;
    if (lin_solver) a.LinSolver = lin_solver;
    a.LinSolverTolerance = tol; a.LinSolverIterMax = max_iter; a.IncNumb = inc_numb;
    d->db.has_analysis = true;
    return STAN_HOST_OK;
}

int stan_host_db_sizes(stan_db *d, int64_t s[8]) {
    if (!d || !s) return STAN_HOST_E_ARG;
    s[0] = (int64_t)d->db.NodeLib.Count(); s[1] = (int64_t)d->db.ElemLib.Count();
    s[2] = (int64_t)d->db.MatLib.Count(); s[3] = (int64_t)d->db.BCLib.Count();
    s[4] = d->db.nDOF; s[5] = d->db.AnalysisLib.Result_StepNo;
    s[6] = (int64_t)d->db.Import_Error.size(); s[7] = 0;
    return STAN_HOST_OK;
}

int stan_host_db_get_analysis(stan_db *d, char *type, char *lin_solver, int32_t cap, double *tol,
                              int32_t *max_iter, int32_t *result_step) {
    if (!d) return STAN_HOST_E_ARG;
    const Analysis &a = d->db.AnalysisLib;
    if (type && cap > 0) { strncpy(type, a.Type.c_str(), (size_t)cap - 1); type[cap - 1] = 0; }
    if (lin_solver && cap > 0) { strncpy(lin_solver, a.LinSolver.c_str(), (size_t)cap - 1); lin_solver[cap - 1] = 0; }
    if (tol) *tol = a.LinSolverTolerance;
    if (max_iter) *max_iter = a.LinSolverIterMax;
    if (result_step) *result_step = a.Result_StepNo;
    return STAN_HOST_OK;
}

int stan_host_db_assign_dof(stan_db *d) {
    if (!d) return STAN_HOST_E_ARG;
    const int rc = d->db.AssignDOF();
    if (rc) d->err = "AssignDOF failed (" + std::to_string(rc) + ")";
    return rc;
}

int stan_host_db_get_flat(stan_db *d, double *xyz, int32_t *node_ids, int32_t *node_dof,
                          int32_t *conn, int32_t *elem_ids, int32_t *elem_mat,
                          uint8_t *elem_type, double *mat_E_nu, int32_t cap_mat, int32_t *n_mat) {
    if (!d) return STAN_HOST_E_ARG;
    FlatModel f;
    const int rc = Flatten(d->db, &f, &d->err);
    if (rc) return rc;
    if (xyz) memcpy(xyz, f.xyz.data(), f.xyz.size() * 8);
    if (node_dof) memcpy(node_dof, f.node_dof.data(), f.node_dof.size() * 4);
    if (conn) memcpy(conn, f.conn.data(), f.conn.size() * 4);
    if (elem_mat) memcpy(elem_mat, f.elem_mat.data(), f.elem_mat.size() * 4);
    if (elem_type) memcpy(elem_type, f.elem_type.data(), f.elem_type.size());
    if (node_ids) { size_t i = 0; for (const auto &kv : d->db.NodeLib.Items()) node_ids[i++] = kv.first; }
    if (elem_ids) { size_t i = 0; for (const auto &kv : d->db.ElemLib.Items()) elem_ids[i++] = kv.first; }
    const int32_t nm = (int32_t)(f.mat_E_nu.size() / 2);
    if (n_mat) *n_mat = nm;
    if (mat_E_nu) {
        if (cap_mat < nm) { d->err = "mat_E_nu buffer too small"; return STAN_HOST_E_ARG; }
        memcpy(mat_E_nu, f.mat_E_nu.data(), f.mat_E_nu.size() * 8);
    }
    return STAN_HOST_OK;
}

int stan_host_db_get_reduction(stan_db *d, int32_t *red, int64_t *n_fixed, double *F) {
    if (!d || !n_fixed) return STAN_HOST_E_ARG;
    std::vector<int32_t> r;
    std::vector<double> f;
    const int rc = BuildReductionAndLoads(d->db, &r, n_fixed, &f, &d->err);
    if (rc) return rc;
    if (red) memcpy(red, r.data(), r.size() * 4);
    if (F) memcpy(F, f.data(), f.size() * 8);
    return STAN_HOST_OK;
}

int stan_host_db_set_results(stan_db *d, const double *disp, const double *strain,
                             const double *stress) {
    if (!d || !disp) return STAN_HOST_E_ARG;
    const int inc = 1;
    size_t i = 0;
    for (auto &kv : d->db.NodeLib.Items()) {
        Node &n = kv.second;
        n.Initialize_StepZero();      // Solver.cs:81-85
        n.Initialize_NewDisp(inc);
        n.dU_buffer[0] = disp[3 * i]; n.dU_buffer[1] = disp[3 * i + 1]; n.dU_buffer[2] = disp[3 * i + 2];
        n.Update_Displacement(inc);   // Solver.cs:203-206
        i++;
    }
    i = 0;
    for (auto &kv : d->db.ElemLib.Items()) {
        Element &e = kv.second;
        e.Initialize_StepZero();      // Solver.cs:86-90
        e.Initialize_Increment(inc);
        if (strain && stress && e.NList.size() == 8) {  // Update_StrainStress, Element.cs:257-267
            memcpy(e.Strain[1].M.data(), strain + 48 * i, 48 * 8);
            memcpy(e.Stress[1].M.data(), stress + 48 * i, 48 * 8);
        }
        i++;
    }
    d->db.AnalysisLib.Result_StepNo = 1;  // Solver.cs:56
    return STAN_HOST_OK;
}

int stan_host_db_get_results(stan_db *d, int32_t inc, double *disp, double *strain, double *stress) {
    if (!d || inc < 0) return STAN_HOST_E_ARG;
    size_t i = 0;
    if (disp)
        for (const auto &kv : d->db.NodeLib.Items()) {
            const Node &n = kv.second;
            if (n.DispX.size() <= (size_t)inc || n.DispY.size() <= (size_t)inc || n.DispZ.size() <= (size_t)inc) {
                d->err = "node " + std::to_string(n.ID) + " has no displacement for increment " + std::to_string(inc);
                return STAN_HOST_E_ARG;
            }
            disp[3 * i] = n.DispX[(size_t)inc]; disp[3 * i + 1] = n.DispY[(size_t)inc]; disp[3 * i + 2] = n.DispZ[(size_t)inc];
            i++;
        }
    i = 0;
    if (strain || stress)
        for (const auto &kv : d->db.ElemLib.Items()) {
            const Element &e = kv.second;
            if (e.Strain.size() <= (size_t)inc || e.Stress.size() <= (size_t)inc ||
                e.Strain[(size_t)inc].M.size() != 48 || e.Stress[(size_t)inc].M.size() != 48) {
                d->err = "element " + std::to_string(e.ID) + " has no 8x6 results for increment " + std::to_string(inc);
                return STAN_HOST_E_ARG;
            }
            if (strain) memcpy(strain + 48 * i, e.Strain[(size_t)inc].M.data(), 48 * 8);
            if (stress) memcpy(stress + 48 * i, e.Stress[(size_t)inc].M.data(), 48 * 8);
            i++;
        }
    return STAN_HOST_OK;
}

}  // extern "C"
