// model.cpp -- behaviour of the STAN_Database mirror classes on the linear-static path.
#include "model.h"

#include <atomic>
#include <cstring>
#include <thread>

#include "../../include/stan_host.h"

namespace stan {

void Node::Initialize_StepZero() {  // Node.cs:95-103
    DispX = {0.0}; DispY = {0.0}; DispZ = {0.0};
    dU_buffer[0] = dU_buffer[1] = dU_buffer[2] = 0.0;
}
void Node::Initialize_NewDisp(int inc) {  // Node.cs:109-116
    DispX.push_back(DispX[(size_t)inc - 1]);
    DispY.push_back(DispY[(size_t)inc - 1]);
    DispZ.push_back(DispZ[(size_t)inc - 1]);
    dU_buffer[0] = dU_buffer[1] = dU_buffer[2] = 0.0;
}
void Node::Update_Displacement(int inc) {  // Node.cs:176-181
    DispX[(size_t)inc] += dU_buffer[0];
    DispY[(size_t)inc] += dU_buffer[1];
    DispZ[(size_t)inc] += dU_buffer[2];
}
void Element::Initialize_StepZero() {  // Element.cs:79-87
    Strain.assign(1, MatrixST((int)NList.size(), 6));
    Stress.assign(1, MatrixST((int)NList.size(), 6));
}
void Element::Initialize_Increment(int) {  // Element.cs:92-98
    Strain.emplace_back((int)NList.size(), 6);
    Stress.emplace_back((int)NList.size(), 6);
}

static std::string pad_right(std::string s, size_t w) { if (s.size() < w) s.append(w - s.size(), ' '); return s; }
static std::string pad_left(std::string s, size_t w) { if (s.size() < w) s.insert(0, w - s.size(), ' '); return s; }

std::string Database::Database_Summary() const {  // Database.cs:123-133
    std::string s;
    s += "\n  ==================   DATABASE SUMMARY   ==================";
    s += pad_right("\n   Number of nodes:", 25) + pad_left(std::to_string(NodeLib.Count()), 31);
    s += pad_right("\n   Number of elements:", 25) + pad_left(std::to_string(ElemLib.Count()), 31);
    s += pad_right("\n   Number of DoF:", 25) + pad_left(std::to_string(nDOF), 31);
    s += "\n  ========================================================== \n";
    return s;
}

// Database.cs:140-234.  EList is rebuilt exactly as the reference does (Initialize_EList,
// AddElem2Nodes in ElemLib order, Distinct) because ExportOutput serializes it afterwards;
// the BFS itself is stan_host_assign_dof (dof.cpp).  An NList entry that is not a NodeLib
// key makes the C# throw KeyNotFoundException in AddElem2Nodes.
int Database::AssignDOF() {
    const int64_t nn = (int64_t)NodeLib.Count(), ne = (int64_t)ElemLib.Count();
    // node IDs -> NodeLib positions on the host threads (26 M hash lookups at 148^3); EList in element order after it
    std::vector<int32_t> conn((size_t)ne * 8);
    auto &nodes = NodeLib.Items();
    const auto &elems = ElemLib.Items();
    std::atomic<int> bad{0};
    {
        const size_t nt = (size_t)HostThreads();
        auto work = [&](size_t a, size_t b) {
            for (size_t e = a; e < b; e++) {
                const Element &el = elems[e].second;
                if (el.NList.size() != 8) { bad.store(1); return; }
                for (int k = 0; k < 8; k++) {
                    const int64_t idx = NodeLib.IndexOf(el.NList[(size_t)k]);
                    if (idx < 0) { bad.store(1); return; }
                    conn[e * 8 + (size_t)k] = (int32_t)idx;
                }
            }
        };
        if (nt <= 1 || elems.size() < 4096) work(0, elems.size());
        else {
            std::vector<std::thread> th;
            for (size_t t = 0; t < nt; t++) th.emplace_back(work, elems.size() * t / nt, elems.size() * (t + 1) / nt);
            for (std::thread &x : th) x.join();
        }
        if (bad.load()) return STAN_HOST_E_ARG;
    }
    // the walk, and with it the node -> element incidence it builds anyway (CSR, ElemLib order)
    std::vector<int32_t> index((size_t)nn);
    std::vector<int64_t> eptr;
    std::vector<int32_t> elist;
    const int rc = AssignDofCore(nn, ne, conn.data(), index.data(), nullptr, &eptr, &elist);
    if (rc != STAN_HOST_OK) return rc;
    {   // Element.AddElem2Nodes in ElemLib order (Element.cs:474-480) and Node.SetDOF (Node.cs:218-223), per node on the host threads
        const size_t nt = (size_t)HostThreads();
        auto fill = [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) {
                Node &n = nodes[i].second;
                n.EList.clear();
                n.EList.reserve((size_t)(eptr[i + 1] - eptr[i]));
                for (int64_t q = eptr[i]; q < eptr[i + 1]; q++) n.EList.push_back(elems[(size_t)elist[(size_t)q]].second.ID);
                n.SetDOF(index[i]);
            }
        };
        if (nt <= 1 || nodes.size() < 4096) fill(0, nodes.size());
        else {
            std::vector<std::thread> th;
            for (size_t t = 0; t < nt; t++) th.emplace_back(fill, nodes.size() * t / nt, nodes.size() * (t + 1) / nt);
            for (std::thread &x : th) x.join();
        }
    }
    conn_index = std::move(conn);   // Flatten needs the same 8 lookups per element again
    return STAN_HOST_OK;
}

int Flatten(const Database &db, FlatModel *out, std::string *err) {
    const size_t nn = db.NodeLib.Count(), ne = db.ElemLib.Count();
    out->xyz.resize(nn * 3);
    out->node_dof.resize(nn * 3);
    size_t i = 0;
    for (const auto &kv : db.NodeLib.Items()) {
        const Node &n = kv.second;
        out->xyz[3 * i] = n.X; out->xyz[3 * i + 1] = n.Y; out->xyz[3 * i + 2] = n.Z;
        if (n.DOF.size() != 3) { if (err) *err = "node " + std::to_string(n.ID) + " has no DOF (run AssignDOF)"; return STAN_HOST_E_ARG; }
        for (int d = 0; d < 3; d++) out->node_dof[3 * i + d] = n.DOF[(size_t)d];
        i++;
    }
    // Solver.cs:33-39: only materials whose Type contains "Elastic" get an elastic matrix
    std::unordered_map<int, int> mat_index;
    out->mat_E_nu.clear();
    for (const auto &kv : db.MatLib.Items()) {
        const Material &m = kv.second;
        if (!m.has_type) { if (err) *err = "material " + std::to_string(kv.first) + " has a null Type (Solver.cs:35 would throw)"; return STAN_HOST_E_ARG; }
        if (m.Type.find("Elastic") == std::string::npos) { mat_index[kv.first] = -1; continue; }
        mat_index[kv.first] = (int)(out->mat_E_nu.size() / 2);
        out->mat_E_nu.push_back(m.E);
        out->mat_E_nu.push_back(m.Poisson);
    }
    out->conn.resize(ne * 8);
    out->elem_mat.resize(ne);
    out->elem_type.resize(ne);
    const bool have_conn = db.conn_index.size() == ne * 8;   // AssignDOF resolved the node IDs already
    if (have_conn) out->conn = db.conn_index;
    i = 0;
    for (const auto &kv : db.ElemLib.Items()) {
        const Element &e = kv.second;
        if (e.NList.size() != 8) { if (err) *err = "element " + std::to_string(e.ID) + " does not have 8 nodes"; return STAN_HOST_E_ARG; }
        for (int a = 0; a < 8 && !have_conn; a++) {
            const int64_t idx = db.NodeLib.IndexOf(e.NList[(size_t)a]);
            if (idx < 0) { if (err) *err = "element " + std::to_string(e.ID) + " references unknown node " + std::to_string(e.NList[(size_t)a]); return STAN_HOST_E_ARG; }
            out->conn[8 * i + a] = (int32_t)idx;
        }
        auto mi = mat_index.find(e.MatID);
        if (mi == mat_index.end()) { if (err) *err = "element " + std::to_string(e.ID) + ": MatID " + std::to_string(e.MatID) + " not in MatLib (Element.cs:147 KeyNotFound)"; return STAN_HOST_E_ARG; }
        if (mi->second < 0) { if (err) *err = "element " + std::to_string(e.ID) + ": material " + std::to_string(e.MatID) + " is not Elastic (null ElasticMatrix)"; return STAN_HOST_E_ARG; }
        out->elem_mat[i] = mi->second;
        if (e.Type == "HEX8_G2") out->elem_type[i] = 2;       // FE_Library.cs:45
        else if (e.Type == "HEX8_G1") out->elem_type[i] = 1;  // FE_Library.cs:44
        else { if (err) *err = "element " + std::to_string(e.ID) + ": unsupported Type '" + e.Type + "'"; return STAN_HOST_E_ARG; }
        i++;
    }
    return STAN_HOST_OK;
}

// Solver.cs:104-152
int BuildReductionAndLoads(const Database &db, std::vector<int32_t> *red, int64_t *n_fixed,
                           std::vector<double> *F, std::string *err) {
    const int64_t ndof = db.nDOF;  // the solver trusts the file's nDOF (Solver.cs:121)
    red->assign((size_t)ndof, 0);
    auto node_dofs = [&](int nid, const int **dofs) -> bool {
        const Node *n = db.NodeLib.Find(nid);
        if (!n || n->DOF.size() != 3) return false;
        *dofs = n->DOF.data();
        return true;
    };
    for (const auto &kv : db.BCLib.Items()) {
        const BoundaryCondition &bc = kv.second;
        if (bc.Type != "SPC") continue;  // BCLib.Values.Where(x => x.Type == "SPC")
        for (const auto &nv : bc.NodalValues.Items()) {
            const int *dofs;
            if (!node_dofs(nv.first, &dofs)) { if (err) *err = "SPC on unknown node " + std::to_string(nv.first); return STAN_HOST_E_ARG; }
            if (nv.second.M.size() < 3) { if (err) *err = "SPC value is not 3x1"; return STAN_HOST_E_ARG; }
            for (int d = 0; d < 3; d++)
                if (nv.second.M[(size_t)d] == 1) {  // Solver.cs:110-112
                    if (dofs[d] < 0 || dofs[d] >= ndof) { if (err) *err = "DOF outside nDOF"; return STAN_HOST_E_ARG; }
                    (*red)[(size_t)dofs[d]] = -1;
                }
        }
    }
    int32_t reduc = 0;
    for (int64_t i = 0; i < ndof; i++) {
        if ((*red)[(size_t)i] == -1) reduc++;
        else (*red)[(size_t)i] = reduc;
    }
    *n_fixed = reduc;
    F->assign((size_t)(ndof - reduc), 0.0);
    for (const auto &kv : db.BCLib.Items()) {
        const BoundaryCondition &bc = kv.second;
        if (bc.Type != "PointLoad") continue;
        for (const auto &nv : bc.NodalValues.Items()) {
            const int *dofs;
            if (!node_dofs(nv.first, &dofs)) { if (err) *err = "PointLoad on unknown node " + std::to_string(nv.first); return STAN_HOST_E_ARG; }
            if (nv.second.M.size() < 3) { if (err) *err = "PointLoad value is not 3x1"; return STAN_HOST_E_ARG; }
            for (int dir = 0; dir < 3; dir++) {
                const int dof = dofs[dir];
                if (dof < 0 || dof >= ndof) { if (err) *err = "DOF outside nDOF"; return STAN_HOST_E_ARG; }
                if ((*red)[(size_t)dof] != -1) (*F)[(size_t)(dof - (*red)[(size_t)dof])] += nv.second.M[(size_t)dir];
            }
        }
    }
    return STAN_HOST_OK;
}

}  // namespace stan
