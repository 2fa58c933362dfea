// model.h -- C++ mirror of the STAN_Database object model, as far as the linear-static
// solver path touches it.  Same class and member names as
// the C# DTOs so that the driver (main_solver.cpp) reads like Solver.cs.
//
//   Database          Database.cs:10-21      NodeLib / ElemLib / MatLib / BCLib / nDOF /
//                                            AnalysisLib / Info  ([ProtoMember] 1..7)
//   Node              Node.cs:9-23           ID, X, Y, Z, EList, DOF, DispX/Y/Z
//   Element           Element.cs:11-23       ID, Type, PID, MatID, NList, Strain, Stress
//   MatrixST          MatrixST.cs:15-19      M, Rows, Cols
//   Material          Material.cs:7-14       ID, Type, Name, E, Poisson, ColorID
//   BoundaryCondition BoundaryCondition.cs:8-14  Type, Name, ID, NodalValues, ColorID
//   Analysis          Analysis.cs:6-13       Type, LinSolver, LinSolverTolerance, ...
//   Information/PartInfo  Information.cs:7-40
//
// Dictionaries keep INSERTION order (= STdb wire order), because .NET's
// Dictionary<int,T> enumerates in insertion order when nothing was removed and that
// order drives AssignDOF and the element order.
#pragma once

#include <atomic>
#include <cstdint>
#include <cstring>
#include <functional>
#include <thread>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace stan {

// key -> position index of OrderedDict: open addressing over a power-of-two table (6.5 M keys at 148^3:
// std::unordered_map cost seconds to build and a cache miss per node lookup of every element)
class FlatIndex {
  public:
    void Reserve(size_t n) {
        size_t cap = 16;
        while (cap < 2 * n + 2) cap <<= 1;
        if (cap <= slots_.size()) return;
        std::vector<Slot> old;
        old.swap(slots_);
        slots_.assign(cap, Slot{0, EMPTY});
        shift_ = 64;
        for (size_t c = cap; c > 1; c >>= 1) shift_--;
        for (const Slot &s : old)
            if (s.pos != EMPTY) Put(s.key, s.pos);
    }
    // position of key, or -1
    int64_t Get(int key) const {
        if (slots_.empty()) return -1;
        const size_t mask = slots_.size() - 1;
        for (size_t i = Hash(key);; i = (i + 1) & mask) {
            const Slot &s = slots_[i];
            if (s.pos == EMPTY) return -1;
            if (s.key == key) return (int64_t)s.pos;
        }
    }
    // false when the key exists
    bool Insert(int key, size_t pos) {
        if (2 * (count_ + 1) + 2 > slots_.size()) Reserve(count_ < 8 ? 16 : 2 * count_);
        if (Get(key) >= 0) return false;
        Put(key, (uint32_t)pos);
        count_++;
        return true;
    }
    void Clear() { slots_.clear(); count_ = 0; }
    // Round 5: a whole library at once on `threads` threads (6.5 M keys at 148^3: the serial build was a fifth of the
    // file's read time).  Every key takes a slot of the linear-probing table with a compare-and-swap on the 8-byte slot;
    // a key that is already there keeps the SMALLER position (the first entry in wire order, whichever thread came
    // first).  Returns the number of distinct keys; Get works as after the serial build (probe sequences are the same
    // sets of slots, lookups never depend on insertion order).
    template <typename KeyAt>
    size_t BuildParallel(size_t n, int threads, KeyAt key_at) {
        Clear();
        Reserve(n);
        if (slots_.empty()) return 0;
        std::atomic<size_t> distinct{0};
        const size_t mask = slots_.size() - 1;
        unsigned long long *raw = reinterpret_cast<unsigned long long *>(slots_.data());
        auto pack = [](int key, uint32_t pos) { Slot s{key, pos}; unsigned long long v; memcpy(&v, &s, 8); return v; };
        const unsigned long long empty = pack(0, EMPTY);
        auto work = [&](size_t a, size_t b) {
            size_t mine = 0;
            for (size_t i = a; i < b; i++) {
                const int key = key_at(i);
                for (size_t h = Hash(key);; h = (h + 1) & mask) {
                    unsigned long long cur = __atomic_load_n(raw + h, __ATOMIC_RELAXED);
                    Slot cs; memcpy(&cs, &cur, 8);
                    if (cs.pos == EMPTY) {
                        unsigned long long expect = empty;
                        if (__atomic_compare_exchange_n(raw + h, &expect, pack(key, (uint32_t)i), false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { mine++; break; }
                        cur = expect; memcpy(&cs, &cur, 8);   // somebody took the slot: look at what is there now
                    }
                    if (cs.key == key) {   // a repeated key: the smaller position stays
                        while (cs.pos > (uint32_t)i) {
                            unsigned long long expect = cur;
                            if (__atomic_compare_exchange_n(raw + h, &expect, pack(key, (uint32_t)i), false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
                            cur = expect; memcpy(&cs, &cur, 8);
                        }
                        break;
                    }
                }
            }
            distinct.fetch_add(mine);
        };
        if (threads <= 1 || n < 65536) work(0, n);
        else {
            std::vector<std::thread> th;
            for (int t = 1; t < threads; t++) th.emplace_back(work, n * (size_t)t / (size_t)threads, n * (size_t)(t + 1) / (size_t)threads);
            work(0, n / (size_t)threads);
            for (std::thread &x : th) x.join();
        }
        count_ = distinct.load();
        return count_;
    }

  private:
    static constexpr uint32_t EMPTY = 0xffffffffu;
    struct alignas(8) Slot { int key; uint32_t pos; };
    size_t Hash(int key) const { return (size_t)(((uint64_t)(uint32_t)key * 0x9E3779B97F4A7C15ull) >> shift_); }
    void Put(int key, uint32_t pos) {
        const size_t mask = slots_.size() - 1;
        size_t i = Hash(key);
        while (slots_[i].pos != EMPTY) i = (i + 1) & mask;
        slots_[i] = Slot{key, pos};
    }
    std::vector<Slot> slots_;
    size_t count_ = 0;
    int shift_ = 60;
};

template <typename T>
class OrderedDict {  // Dictionary<int, T> with insertion-order enumeration
  public:
    bool Add(int key, T value) {  // false when the key exists (Dictionary.Add would throw)
        if (!index_.Insert(key, items_.size())) return false;
        items_.emplace_back(key, std::move(value));
        return true;
    }
    // A whole library at once (the parallel STdb reader decodes the entries first): wire order is kept, a
    // repeated key keeps its first entry, like Add.
    void Adopt(std::vector<std::pair<int, T>> &&items, int threads = 1) {
        items_ = std::move(items);
        if (threads > 1 && items_.size() >= 65536) {   // the index on the host threads; no repeated key (any sane file): done
            const std::vector<std::pair<int, T>> &it = items_;
            if (index_.BuildParallel(it.size(), threads, [&it](size_t i) { return it[i].first; }) == it.size()) return;
        }
        index_.Clear();
        index_.Reserve(items_.size());
        size_t w = 0;
        for (size_t i = 0; i < items_.size(); i++) {
            if (!index_.Insert(items_[i].first, w)) continue;   // duplicate key: dropped
            if (w != i) items_[w] = std::move(items_[i]);
            w++;
        }
        items_.resize(w);
    }
    bool ContainsKey(int key) const { return index_.Get(key) >= 0; }
    T *Find(int key) {
        const int64_t i = index_.Get(key);
        return i < 0 ? nullptr : &items_[(size_t)i].second;
    }
    const T *Find(int key) const {
        const int64_t i = index_.Get(key);
        return i < 0 ? nullptr : &items_[(size_t)i].second;
    }
    int64_t IndexOf(int key) const { return index_.Get(key); }
    size_t Count() const { return items_.size(); }
    std::vector<std::pair<int, T>> &Items() { return items_; }
    const std::vector<std::pair<int, T>> &Items() const { return items_; }
    void Clear() { items_.clear(); index_.Clear(); }

  private:
    std::vector<std::pair<int, T>> items_;
    FlatIndex index_;
};

struct MatrixST {  // MatrixST.cs:15-26
    std::vector<double> M;
    int Rows = 0, Cols = 0;
    MatrixST() {}
    MatrixST(int rows, int cols) : M((size_t)rows * cols, 0.0), Rows(rows), Cols(cols) {}
    double Get(int r, int c) const { return M[(size_t)r * Cols + c]; }
    void Set(int r, int c, double v) { M[(size_t)r * Cols + c] = v; }
};

struct Node {  // Node.cs:9-23
    int ID = 0;
    double X = 0, Y = 0, Z = 0;
    std::vector<int> EList;
    std::vector<int> DOF;  // int[3]
    std::vector<double> DispX, DispY, DispZ;
    double dU_buffer[3] = {0, 0, 0};  // not serialized
    // Node(string input): 8-character fixed-field GRID parser, Node.cs:25-80
    static bool FromBdfLine(const std::string &input, Node *out);
    void Initialize_StepZero();        // Node.cs:95-103
    void Initialize_NewDisp(int inc);  // Node.cs:109-116
    void Update_Displacement(int inc); // Node.cs:176-181
    void SetDOF(int index) { DOF = {3 * index, 3 * index + 1, 3 * index + 2}; }  // :218-223
};

struct Element {  // Element.cs:11-23
    int ID = 0;
    std::string Type;
    int PID = 0;
    int MatID = 0;
    std::vector<int> NList;
    std::vector<MatrixST> Strain, Stress;
    bool has_type = false;
    // Element(string input): whitespace split, Element.cs:35-73
    static bool FromBdfLine(const std::string &input, Element *out);
    void Initialize_StepZero();         // Element.cs:79-87
    void Initialize_Increment(int inc); // Element.cs:92-113
};

struct Material {  // Material.cs:7-29
    int ID = 0;
    std::string Type, Name;
    bool has_type = false, has_name = false;
    double E = 0, Poisson = 0;
    int ColorID = 0;
    static Material Create(int id) {  // Material(int id), Material.cs:19-29
        Material m;
        m.ID = id; m.Type = "Elastic"; m.has_type = true; m.ColorID = id % 9;
        m.E = -999; m.Poisson = -999;
        return m;
    }
};

struct BoundaryCondition {  // BoundaryCondition.cs:8-37
    std::string Type, Name;
    bool has_type = false, has_name = false;
    int ID = 0;
    OrderedDict<MatrixST> NodalValues;  // node ID -> 3x1
    int ColorID = 0;
};

struct Analysis {  // Analysis.cs:6-25
    std::string Type = "Linear_Statics", LinSolver = "CG";
    double LinSolverTolerance = 1.0e-6;
    int LinSolverIterMax = 0, IncNumb = 0, Result_StepNo = 0;
};

struct PartInfo {  // Information.cs:33-63
    int ColorID = 0, MatID = 0;
    std::string Name = "blank", HEX_Type = "blank", PENTA_Type = "blank", TET_Type = "blank";
};
struct Information {  // Information.cs:7-30
    OrderedDict<PartInfo> InfoPart;
    bool has_parts = false;
};

struct Database {  // Database.cs:10-37
    OrderedDict<Node> NodeLib;
    OrderedDict<Element> ElemLib;
    OrderedDict<Material> MatLib;
    OrderedDict<BoundaryCondition> BCLib;
    int nDOF = 0;
    Analysis AnalysisLib;
    bool has_analysis = true;
    Information Info;
    bool has_info = true;
    std::vector<std::string> Import_Error;  // not serialized (Database.cs:18)
    std::vector<int32_t> conn_index;        // not serialized: NList as NodeLib positions, left by AssignDOF for Flatten
    // Results of increment 1 as FLAT arrays (not serialized themselves): when set, WriteStdb encodes node i's
    // DispX/Y/Z = {0, disp[3i+d]} and element e's Strain / Stress = {zeros(8x6), strain[48e ..]} -- byte for byte
    // what Initialize_StepZero / Initialize_NewDisp / Update_Displacement / Initialize_Increment /
    // Update_StrainStress leave in the objects (Solver.cs:81-90, 171-178, 203-210), without building 13 million
    // small heap objects first (3.3 s of a 8.9 s run at 148^3).  The pointers must outlive the write.
    // Round 5: the element results may stay on the device until the writer needs them -- `fetch(e0, e1, &strain, &stress)`
    // hands the rows of elements [e0, e1) to the CALLING writer thread ([(e1 - e0) * 48] each, valid until that
    // thread's next call; the console driver forwards it to stan_hip_results_map), so the download overlaps the
    // encoding; strain / stress stay null then.
    struct ResultView {
        const double *disp = nullptr, *strain = nullptr, *stress = nullptr;
        std::function<bool(size_t, size_t, const double **, const double **)> fetch;
        size_t elem_base = 0;   // (writer-internal) element index that strain[0] / stress[0] belong to
    } results;

    // Database.cs:39-111 (mesh only: Part objects are GUI-side and not serialized)
    bool ReadNastranMesh(const std::string &path, std::string *err);
    void Set_nDOF() { nDOF = (int)NodeLib.Count() * 3; }  // Database.cs:135-138
    int AssignDOF();                                       // Database.cs:140-234 (0 / STAN_HOST_E_*)
    std::string Database_Summary() const;                  // Database.cs:123-133
};

// dof.cpp: stan_host_assign_dof with the node -> element incidence (CSR, ElemLib order) it builds handed out as well
int AssignDofCore(int64_t n_nodes, int64_t n_elem, const int32_t *conn, int32_t *node_index_out, int32_t *node_dof_out,
                  std::vector<int64_t> *eptr_out, std::vector<int32_t> *elist_out);

// ---- STdb (protobuf-net 3.0.73 wire format) --------------------------
// packed=false writes repeated scalars unpacked (protobuf-net default without IsPacked);
// the reader accepts both encodings and both orders of map key/value.
bool ReadStdb(const std::string &path, Database *db, std::string *err);
bool ParseStdb(const uint8_t *data, size_t size, Database *db, std::string *err);
bool WriteStdb(const Database &db, const std::string &path, bool packed, std::string *err);
// worker threads of the STdb reader / writer and of the result write-back (STAN_HOST_THREADS, default: the
// machine's cores, at most 16)
int HostThreads();
void SerializeStdb(const Database &db, bool packed, std::string *out);

// ---- flat views for the C-ABI of libstan_hip.so ----------------------------------------------
struct FlatModel {
    std::vector<double> xyz;       // [n_nodes*3] NodeLib order
    std::vector<int32_t> node_dof; // [n_nodes*3]
    std::vector<int32_t> conn;     // [n_elem*8] node indices
    std::vector<int32_t> elem_mat; // [n_elem] index into mat_E_nu
    std::vector<uint8_t> elem_type;
    std::vector<double> mat_E_nu;  // [n_mat*2]
};
// Returns 0, or a negative code with *err: unknown node ID in an NList, MatID not in
// MatLib (the C# throws KeyNotFound, Element.cs:147), material whose Type does not contain
// "Elastic" (ElasticMatrix stays null, Solver.cs:33-39), unsupported element Type,
// element without 8 nodes.
int Flatten(const Database &db, FlatModel *out, std::string *err);

// Solver.cs:104-152: Fix_DOF / nDOF_reduction / F.  Returns 0 or negative.
int BuildReductionAndLoads(const Database &db, std::vector<int32_t> *red, int64_t *n_fixed,
                           std::vector<double> *F, std::string *err);

}  // namespace stan
