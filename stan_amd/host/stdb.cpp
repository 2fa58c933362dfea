// stdb.cpp -- STdb reader/writer: the protobuf wire format protobuf-net 3.0.73 produces for
// the [ProtoContract]/[ProtoMember] attributes of STAN_Database (schema reconstructed in
// SURVEY.md Appendix A; the root object is a bare Database message, no length prefix:
// SolverFunctions.cs:48-63).  protobuf-net is not vendored in the reference and no sample
// .STdb exists in the checkout, so byte-level compatibility with the Windows GUI is
// unverified; the contract tested here is: reader accepts packed and unpacked repeated
// scalars and either order of map key/value, unknown fields are skipped, and
// write(read(bytes)) == bytes for files this writer produced.
//
// Writer rules: fields in ascending number; scalar members equal to 0 / 0.0 / null are
// omitted (implicit zero default); list elements are always written, zeros included;
// empty lists are omitted; repeated scalars unpacked unless `packed`.
//
// Round 3: both directions run on HostThreads() threads.  The libraries are sequences of
// length-delimited map entries, so the reader first walks the top level (tags and lengths only),
// then decodes the entries of a library in parallel and adopts them in wire order; the writer
// encodes chunks of entries in parallel and writes the chunks in order -- the bytes are those of
// the serial writer (tests/test_stdb_pin.py, tests/test_stdb.py).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <thread>

#include <cerrno>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/statvfs.h>
#include <unistd.h>

#include "model.h"

namespace stan {

int HostThreads() {
    if (const char *e = getenv("STAN_HOST_THREADS")) {
        const int n = atoi(e);
        if (n >= 1) return n > 64 ? 64 : n;
    }
    const unsigned hc = std::thread::hardware_concurrency();
    return hc == 0 ? 1 : hc > 16 ? 16 : (int)hc;
}

namespace {

// fn(t) on `threads` threads (t = 0 .. threads-1); the caller is thread 0
template <typename F>
void run_threads(int threads, F fn) {
    std::vector<std::thread> th;
    for (int t = 1; t < threads; t++) th.emplace_back([&fn, t] { fn(t); });
    fn(0);
    for (std::thread &x : th) x.join();
}

// ---------------------------------------------------------------- writer
struct W {
    std::string &o;
    bool packed;
    void varint(uint64_t v) {
        while (v >= 0x80) { o.push_back((char)(v | 0x80)); v >>= 7; }
        o.push_back((char)v);
    }
    void tag(int field, int wt) { varint(((uint64_t)field << 3) | (uint64_t)wt); }
    void i32(int field, int v) {  // implicit zero default
        if (v == 0) return;
        tag(field, 0);
        varint((uint64_t)(int64_t)v);  // negatives: 10-byte two's complement
    }
    void f64raw(double v) {   // fixed64, little endian (the hosts this builds for are)
        o.append(reinterpret_cast<const char *>(&v), 8);
    }
    void f64(int field, double v) {
        if (!(v != 0.0)) return;
        tag(field, 1);
        f64raw(v);
    }
    void str(int field, const std::string &s, bool present) {
        if (!present) return;
        tag(field, 2);
        varint(s.size());
        o += s;
    }
    void bytes(int field, const std::string &s) {
        tag(field, 2);
        varint(s.size());
        o += s;
    }
    void rep_i32(int field, const std::vector<int> &v) {
        if (v.empty()) return;
        if (packed) {
            std::string t;
            W w{t, packed};
            for (int x : v) w.varint((uint64_t)(int64_t)x);
            bytes(field, t);
        } else
            for (int x : v) { tag(field, 0); varint((uint64_t)(int64_t)x); }
    }
    void rep_f64(int field, const std::vector<double> &v) {
        if (v.empty()) return;
        if (packed) {
            tag(field, 2);
            varint(v.size() * 8);
            for (double x : v) f64raw(x);
        } else
            for (double x : v) { tag(field, 1); f64raw(x); }
    }
};

void enc(const MatrixST &m, bool packed, std::string &o) {
    W w{o, packed};
    w.rep_f64(1, m.M);
    w.i32(2, m.Rows);
    w.i32(3, m.Cols);
}
void enc(const Node &n, bool packed, std::string &o) {
    W w{o, packed};
    w.i32(1, n.ID); w.f64(2, n.X); w.f64(3, n.Y); w.f64(4, n.Z);
    w.rep_i32(5, n.EList); w.rep_i32(6, n.DOF);
    w.rep_f64(7, n.DispX); w.rep_f64(8, n.DispY); w.rep_f64(9, n.DispZ);
}
void enc(const Element &e, bool packed, std::string &o) {
    W w{o, packed};
    w.i32(1, e.ID); w.str(2, e.Type, e.has_type); w.i32(3, e.PID); w.i32(4, e.MatID);
    w.rep_i32(5, e.NList);
    thread_local std::string t;   // (one buffer per thread: 4 matrices per element, millions of elements)
    for (const MatrixST &m : e.Strain) { t.clear(); enc(m, packed, t); w.bytes(6, t); }
    for (const MatrixST &m : e.Stress) { t.clear(); enc(m, packed, t); w.bytes(7, t); }
}
void enc(const Material &m, bool packed, std::string &o) {
    W w{o, packed};
    w.i32(1, m.ID); w.str(2, m.Type, m.has_type); w.str(3, m.Name, m.has_name);
    w.f64(4, m.E); w.f64(5, m.Poisson); w.i32(6, m.ColorID);
}
void enc(const PartInfo &p, bool packed, std::string &o);
void enc(const BoundaryCondition &b, bool packed, std::string &o);
template <typename T>
void enc_entry(int key, const T &v, bool packed, std::string &o) {  // map entry {1:key, 2:value}
    W w{o, packed};
    w.i32(1, key);
    std::string t;
    enc(v, packed, t);
    w.bytes(2, t);
}
// node / element with the results of increment 1 taken from flat arrays (Database::ResultView): the same
// bytes as enc() of an object that went through the reference's initialise / update calls
void enc_with_results(const Node &n, size_t i, const Database::ResultView &rv, bool packed, std::string &o) {
    W w{o, packed};
    w.i32(1, n.ID); w.f64(2, n.X); w.f64(3, n.Y); w.f64(4, n.Z);
    w.rep_i32(5, n.EList); w.rep_i32(6, n.DOF);
    thread_local std::vector<double> d(2, 0.0);   // {Disp[0] = 0, Disp[1] = 0 + dU}
    for (int c = 0; c < 3; c++) { d[1] = 0.0 + rv.disp[3 * i + (size_t)c]; w.rep_f64(7 + c, d); }
}
// enc(MatrixST{M = m[0..48), Rows = 8, Cols = 6}) as field `field` of the enclosing message, straight from the row: the
// bytes of w.bytes(field, enc(matrix)) without the matrix object and the two copies (round 5: the export at 148^3 spent
// most of its 1.2 s in byte-wise appends; 6.4 GB of these records)
inline void enc_matrix48_field(int field, const double *m, bool packed, std::string &o) {
    char buf[8 + 48 * 9 + 8];
    char *q = buf;
    const unsigned len = packed ? 3 + 384 + 4 : 48 * 9 + 4;            // 391 / 436: two-byte varints
    *q++ = (char)((field << 3) | 2);
    *q++ = (char)((len & 0x7f) | 0x80); *q++ = (char)(len >> 7);
    if (packed) { *q++ = 0x0A; *q++ = (char)0x80; *q++ = 0x03; memcpy(q, m, 384); q += 384; }   // tag(1, 2), varint(384), raw
    else for (int i = 0; i < 48; i++) { *q++ = 0x09; memcpy(q, m + i, 8); q += 8; }              // tag(1, 1) + fixed64 each
    *q++ = 0x10; *q++ = 8; *q++ = 0x18; *q++ = 6;                                                // Rows = 8, Cols = 6
    o.append(buf, (size_t)(q - buf));
}
void enc_with_results(const Element &e, size_t i, const Database::ResultView &rv, bool packed, std::string &o) {
    W w{o, packed};
    w.i32(1, e.ID); w.str(2, e.Type, e.has_type); w.i32(3, e.PID); w.i32(4, e.MatID);
    w.rep_i32(5, e.NList);
    static const double zero[48] = {0};   // Strain[0] / Stress[0]: what Initialize_StepZero leaves (list elements are written, zeros included)
    enc_matrix48_field(6, zero, packed, o);
    enc_matrix48_field(6, rv.strain + 48 * (i - rv.elem_base), packed, o);
    enc_matrix48_field(7, zero, packed, o);
    enc_matrix48_field(7, rv.stress + 48 * (i - rv.elem_base), packed, o);
}
template <typename T>
void enc_with_results(const T &v, size_t, const Database::ResultView &, bool packed, std::string &o) { enc(v, packed, o); }
inline bool has_results(const Node &, const Database::ResultView *rv) { return rv && rv->disp; }
inline bool has_results(const Element &e, const Database::ResultView *rv) { return rv && rv->strain && rv->stress && e.NList.size() == 8; }
// the element rows of a chunk from wherever they live (Database::ResultView::fetch): a view of [i0, i1) for this thread
inline bool fetch_rows(const std::vector<std::pair<int, Element>> &, size_t i0, size_t i1, const Database::ResultView *rv,
                       Database::ResultView *local) {
    if (!rv || !rv->fetch || rv->strain) return true;
    *local = *rv;
    local->fetch = nullptr;
    local->elem_base = i0;
    return rv->fetch(i0, i1, &local->strain, &local->stress);
}
template <typename T>
inline bool fetch_rows(const std::vector<std::pair<int, T>> &, size_t, size_t, const Database::ResultView *, Database::ResultView *) { return true; }
template <typename T> inline bool has_results(const T &, const Database::ResultView *) { return false; }

// field `field` of the root message for entries [i0, i1) of a library, appended to o
template <typename T>
bool enc_lib_range(int field, const std::vector<std::pair<int, T>> &items, size_t i0, size_t i1, bool packed,
                   std::string &o, const Database::ResultView *rv = nullptr) {
    Database::ResultView local;
    if (!fetch_rows(items, i0, i1, rv, &local)) return false;
    if (local.strain) rv = &local;
    W w{o, packed};
    std::string entry, val;
    for (size_t i = i0; i < i1; i++) {
        entry.clear(); val.clear();
        W we{entry, packed};
        we.i32(1, items[i].first);
        if (has_results(items[i].second, rv)) enc_with_results(items[i].second, i, *rv, packed, val);
        else enc(items[i].second, packed, val);
        we.bytes(2, val);
        w.bytes(field, entry);
    }
    return true;
}
void enc(const BoundaryCondition &b, bool packed, std::string &o) {
    W w{o, packed};
    w.str(1, b.Type, b.has_type); w.str(2, b.Name, b.has_name); w.i32(3, b.ID);
    for (const auto &kv : b.NodalValues.Items()) {
        std::string t;
        enc_entry(kv.first, kv.second, packed, t);
        w.bytes(4, t);
    }
    w.i32(5, b.ColorID);
}
void enc(const Analysis &a, bool packed, std::string &o) {
    W w{o, packed};
    w.str(1, a.Type, true); w.str(2, a.LinSolver, true); w.f64(3, a.LinSolverTolerance);
    w.i32(4, a.LinSolverIterMax); w.i32(5, a.IncNumb); w.i32(6, a.Result_StepNo);
}
void enc(const PartInfo &p, bool packed, std::string &o) {
    W w{o, packed};
    w.i32(1, p.ColorID); w.i32(2, p.MatID); w.str(3, p.Name, true); w.str(4, p.HEX_Type, true);
    w.str(5, p.PENTA_Type, true); w.str(6, p.TET_Type, true);
}
void enc(const Information &i, bool packed, std::string &o) {
    W w{o, packed};
    for (const auto &kv : i.InfoPart.Items()) {
        std::string t;
        enc_entry(kv.first, kv.second, packed, t);
        w.bytes(1, t);
    }
}

// ---------------------------------------------------------------- reader
struct R {
    const uint8_t *p, *end;
    bool ok = true;
    bool more() const { return ok && p < end; }
    uint64_t varint() {
        uint64_t v = 0;
        for (int s = 0; s < 70; s += 7) {
            if (p >= end) { ok = false; return 0; }
            const uint8_t b = *p++;
            v |= (uint64_t)(b & 0x7f) << s;
            if (!(b & 0x80)) return v;
        }
        ok = false;
        return 0;
    }
    double f64() {
        if (end - p < 8) { ok = false; return 0; }
        uint64_t b = 0;
        for (int i = 0; i < 8; i++) b |= (uint64_t)p[i] << (8 * i);
        p += 8;
        double d;
        memcpy(&d, &b, 8);
        return d;
    }
    R sub() {  // length-delimited payload
        const uint64_t n = varint();
        if (!ok || n > (uint64_t)(end - p)) { ok = false; return R{p, p, false}; }
        R r{p, p + n, true};
        p += n;
        return r;
    }
    void skip(int wt) {
        switch (wt) {
            case 0: varint(); break;
            case 1: if (end - p < 8) ok = false; else p += 8; break;
            case 2: sub(); break;
            case 5: if (end - p < 4) ok = false; else p += 4; break;
            default: ok = false;
        }
    }
    std::string str() {
        R s = sub();
        return s.ok ? std::string((const char *)s.p, (size_t)(s.end - s.p)) : std::string();
    }
    // (unpacked lists arrive one element per field: room for the usual sizes at the first one -- NList / EList 8, DOF 3,
    // Disp 2 -- instead of the 1, 2, 4, 8 growth: ten small allocations per node were a third of the read at 148^3)
    void rep_i32(int wt, std::vector<int> &v) {
        if (wt == 0) { if (v.capacity() == 0) v.reserve(8); v.push_back((int)(int64_t)varint()); }
        else if (wt == 2) { R s = sub(); while (s.more()) v.push_back((int)(int64_t)s.varint()); ok &= s.ok; }
        else skip(wt);
    }
    void rep_f64(int wt, std::vector<double> &v) {
        if (wt == 1) { if (v.capacity() == 0) v.reserve(2); v.push_back(f64()); }
        else if (wt == 2) { R s = sub(); while (s.more()) v.push_back(s.f64()); ok &= s.ok; }
        else skip(wt);
    }
};

#define FIELDS(r)                                 \
    while ((r).more()) {                          \
        const uint64_t tg_ = (r).varint();        \
        if (!(r).ok) break;                       \
        const int f = (int)(tg_ >> 3), wt = (int)(tg_ & 7);
#define END_FIELDS(r) }

bool dec(R r, MatrixST &m) {
    FIELDS(r)
        if (f == 1) r.rep_f64(wt, m.M);
        else if (f == 2 && wt == 0) m.Rows = (int)r.varint();
        else if (f == 3 && wt == 0) m.Cols = (int)r.varint();
        else r.skip(wt);
    END_FIELDS(r)
    return r.ok;
}
bool dec(R r, Node &n) {
    FIELDS(r)
        if (f == 1 && wt == 0) n.ID = (int)r.varint();
        else if (f == 2 && wt == 1) n.X = r.f64();
        else if (f == 3 && wt == 1) n.Y = r.f64();
        else if (f == 4 && wt == 1) n.Z = r.f64();
        else if (f == 5) r.rep_i32(wt, n.EList);
        else if (f == 6) r.rep_i32(wt, n.DOF);
        else if (f == 7) r.rep_f64(wt, n.DispX);
        else if (f == 8) r.rep_f64(wt, n.DispY);
        else if (f == 9) r.rep_f64(wt, n.DispZ);
        else r.skip(wt);
    END_FIELDS(r)
    return r.ok;
}
bool dec(R r, Element &e) {
    FIELDS(r)
        if (f == 1 && wt == 0) e.ID = (int)r.varint();
        else if (f == 2 && wt == 2) { e.Type = r.str(); e.has_type = true; }
        else if (f == 3 && wt == 0) e.PID = (int)r.varint();
        else if (f == 4 && wt == 0) e.MatID = (int)r.varint();
        else if (f == 5) r.rep_i32(wt, e.NList);
        else if (f == 6 && wt == 2) { e.Strain.emplace_back(); if (!dec(r.sub(), e.Strain.back())) r.ok = false; }
        else if (f == 7 && wt == 2) { e.Stress.emplace_back(); if (!dec(r.sub(), e.Stress.back())) r.ok = false; }
        else r.skip(wt);
    END_FIELDS(r)
    return r.ok;
}
bool dec(R r, Material &m) {
    FIELDS(r)
        if (f == 1 && wt == 0) m.ID = (int)r.varint();
        else if (f == 2 && wt == 2) { m.Type = r.str(); m.has_type = true; }
        else if (f == 3 && wt == 2) { m.Name = r.str(); m.has_name = true; }
        else if (f == 4 && wt == 1) m.E = r.f64();
        else if (f == 5 && wt == 1) m.Poisson = r.f64();
        else if (f == 6 && wt == 0) m.ColorID = (int)r.varint();
        else r.skip(wt);
    END_FIELDS(r)
    return r.ok;
}
template <typename T>
bool dec_entry(R r, int &key, T &val) {  // key and value in either order
    key = 0;
    FIELDS(r)
        if (f == 1 && wt == 0) key = (int)(int64_t)r.varint();
        else if (f == 2 && wt == 2) { if (!dec(r.sub(), val)) r.ok = false; }
        else r.skip(wt);
    END_FIELDS(r)
    return r.ok;
}
bool dec(R r, BoundaryCondition &b) {
    FIELDS(r)
        if (f == 1 && wt == 2) { b.Type = r.str(); b.has_type = true; }
        else if (f == 2 && wt == 2) { b.Name = r.str(); b.has_name = true; }
        else if (f == 3 && wt == 0) b.ID = (int)r.varint();
        else if (f == 4 && wt == 2) {
            int k; MatrixST m;
            if (!dec_entry(r.sub(), k, m)) r.ok = false;
            else b.NodalValues.Add(k, std::move(m));
        }
        else if (f == 5 && wt == 0) b.ColorID = (int)r.varint();
        else r.skip(wt);
    END_FIELDS(r)
    return r.ok;
}
bool dec(R r, Analysis &a) {
    // SkipConstructor = true: absent members are null / 0, not the constructor defaults
    a = Analysis();
    a.Type.clear(); a.LinSolver.clear(); a.LinSolverTolerance = 0;
    FIELDS(r)
        if (f == 1 && wt == 2) a.Type = r.str();
        else if (f == 2 && wt == 2) a.LinSolver = r.str();
        else if (f == 3 && wt == 1) a.LinSolverTolerance = r.f64();
        else if (f == 4 && wt == 0) a.LinSolverIterMax = (int)r.varint();
        else if (f == 5 && wt == 0) a.IncNumb = (int)r.varint();
        else if (f == 6 && wt == 0) a.Result_StepNo = (int)r.varint();
        else r.skip(wt);
    END_FIELDS(r)
    return r.ok;
}
bool dec(R r, PartInfo &p) {
    FIELDS(r)
        if (f == 1 && wt == 0) p.ColorID = (int)r.varint();
        else if (f == 2 && wt == 0) p.MatID = (int)r.varint();
        else if (f == 3 && wt == 2) p.Name = r.str();
        else if (f == 4 && wt == 2) p.HEX_Type = r.str();
        else if (f == 5 && wt == 2) p.PENTA_Type = r.str();
        else if (f == 6 && wt == 2) p.TET_Type = r.str();
        else r.skip(wt);
    END_FIELDS(r)
    return r.ok;
}
bool dec(R r, Information &i) {
    FIELDS(r)
        if (f == 1 && wt == 2) {
            int k; PartInfo p;
            if (!dec_entry(r.sub(), k, p)) r.ok = false;
            else { i.InfoPart.Add(k, std::move(p)); i.has_parts = true; }
        }
        else r.skip(wt);
    END_FIELDS(r)
    return r.ok;
}

}  // namespace

void SerializeStdb(const Database &db, bool packed, std::string *out) {
    std::string &o = *out;
    W w{o, packed};
    for (const auto &kv : db.NodeLib.Items()) { std::string t; enc_entry(kv.first, kv.second, packed, t); w.bytes(1, t); }
    for (const auto &kv : db.ElemLib.Items()) { std::string t; enc_entry(kv.first, kv.second, packed, t); w.bytes(2, t); }
    for (const auto &kv : db.MatLib.Items()) { std::string t; enc_entry(kv.first, kv.second, packed, t); w.bytes(3, t); }
    for (const auto &kv : db.BCLib.Items()) { std::string t; enc_entry(kv.first, kv.second, packed, t); w.bytes(4, t); }
    w.i32(5, db.nDOF);
    if (db.has_analysis) { std::string t; enc(db.AnalysisLib, packed, t); w.bytes(6, t); }
    if (db.has_info) { std::string t; enc(db.Info, packed, t); w.bytes(7, t); }
}

namespace {
// Where the encoded chunks go.  pwrite (default): every chunk at its own offset by the thread that encoded it.  Buffered
// writes to ONE file take the inode's lock one at a time whatever the number of threads, so the export of a large model
// runs at what one writer reaches (measured at 148^3: 6.4 GB in 1.17-1.33 s with 1 or 16 writers, 5.4 GB/s) -- the
// encoding is hidden behind it.  map (STAN_STDB_WRITE=map, kept for file systems where it pays): the file gets an
// upper-bound length, is mapped shared, the chunks are copied into the mapping (page faults instead of write calls) and
// the file is cut to its true length at the end.  On the GPU boxes of round 5 that was SLOWER: 2.5-3.8 s for the same 6.4 GB
// (a write fault per 4-KB page through the file system; profiles/r05/console_driver_148.md).
struct Sink {
    int fd = -1;
    char *map = nullptr;
    size_t map_len = 0;
    bool sequential = false;   // the target cannot seek (a FIFO, /dev/stdout into a pipe, a socket): plain write calls, ONE writer, in order
    bool put(const char *p, size_t n, int64_t off) const {
        if (sequential) {
            while (n > 0) {
                const ssize_t k = write(fd, p, n);
                if (k < 0) { if (errno == EINTR) continue; return false; }
                p += k; n -= (size_t)k;
            }
            return true;
        }
        if (map) {
            if (off < 0 || (size_t)off + n > map_len) return false;   // (the bound is an upper bound: never)
            memcpy(map + off, p, n);
            return true;
        }
        while (n > 0) {
            const ssize_t k = pwrite(fd, p, n, (off_t)off);
            if (k < 0) { if (errno == EINTR) continue; return false; }
            p += k; n -= (size_t)k; off += k;
        }
        return true;
    }
};
// upper bounds of the encoded size of one library entry (tag + varint <= 11 B per scalar, 10 B per list element)
inline size_t max_entry_size(const Node &n, const Database::ResultView *rv) {
    return 96 + 11 * (n.EList.size() + n.DOF.size()) + 10 * (n.DispX.size() + n.DispY.size() + n.DispZ.size()) + (rv ? 3 * 32 : 0);
}
inline size_t max_entry_size(const Element &e, const Database::ResultView *rv) {
    size_t s = 96 + e.Type.size() + 11 * e.NList.size() + (rv ? 4 * 448 : 0);
    for (const MatrixST &m : e.Strain) s += 40 + 10 * m.M.size();
    for (const MatrixST &m : e.Stress) s += 40 + 10 * m.M.size();
    return s;
}
template <typename T>
size_t max_lib_size(const std::vector<std::pair<int, T>> &items, const Database::ResultView *rv, int threads) {
    std::atomic<size_t> total{0};
    auto work = [&](int t) {
        size_t s = 0;
        for (size_t i = items.size() * (size_t)t / (size_t)threads; i < items.size() * (size_t)(t + 1) / (size_t)threads; i++)
            s += max_entry_size(items[i].second, rv);
        total.fetch_add(s);
    };
    if (threads <= 1 || items.size() < 65536) { threads = 1; work(0); }
    else run_threads(threads, work);
    return total.load();
}
// One library to the file at *offset (advanced past it).  Round 5: the `threads` workers encode chunks of entries AND
// write them -- a chunk's place in the file is the sum of the sizes of the chunks before it, known as soon as those are
// ENCODED (not written), so the writes of different chunks overlap (round 3-4: one thread wrote every byte, 4 GB/s, most
// of the export at 148^3).  A worker holds one encoded chunk at a time: bounded memory whatever the model size.  The
// bytes are SerializeStdb's.
template <typename T>
bool write_lib(const Sink &sink, int64_t *offset, int field, const std::vector<std::pair<int, T>> &items, bool packed, int threads,
               const Database::ResultView *rv = nullptr) {
    const size_t n = items.size();
    if (n == 0) return true;
    constexpr size_t CHUNK = 4096;
    const size_t nchunks = (n + CHUNK - 1) / CHUNK;
    if (threads <= 1 || nchunks < 4) {
        std::string buf;
        for (size_t c = 0; c < nchunks; c++) {
            buf.clear();
            if (!enc_lib_range(field, items, c * CHUNK, std::min(n, (c + 1) * CHUNK), packed, buf, rv)) return false;
            if (!sink.put(buf.data(), buf.size(), *offset)) return false;
            *offset += (int64_t)buf.size();
        }
        return true;
    }
    std::vector<int64_t> size(nchunks, -1), off(nchunks + 1, -1);
    off[0] = *offset;
    size_t resolved = 0;     // off[0 .. resolved] are known (guarded by m)
    std::mutex m;
    std::condition_variable cv;
    std::atomic<size_t> next{0};
    std::atomic<bool> ok{true};
    auto worker = [&] {
        std::string buf;
        for (;;) {
            const size_t c = next.fetch_add(1);
            if (c >= nchunks || !ok.load()) return;
            buf.clear();
            const bool enc_ok = enc_lib_range(field, items, c * CHUNK, std::min(n, (c + 1) * CHUNK), packed, buf, rv);
            int64_t at = -1;
            {
                std::unique_lock<std::mutex> lk(m);
                if (!enc_ok) ok.store(false);
                size[c] = (int64_t)buf.size();
                while (resolved < nchunks && size[resolved] >= 0) { off[resolved + 1] = off[resolved] + size[resolved]; resolved++; }
                cv.notify_all();
                // chunks are claimed in ascending order: every earlier chunk is being encoded by another worker
                cv.wait(lk, [&] { return off[c] >= 0 || !ok.load(); });
                at = off[c];
            }
            if (!ok.load()) return;
            if (!sink.put(buf.data(), buf.size(), at)) { ok.store(false); std::lock_guard<std::mutex> lk(m); cv.notify_all(); return; }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; t++) th.emplace_back(worker);
    worker();
    for (std::thread &x : th) x.join();
    if (!ok.load() || resolved != nchunks) return false;
    *offset = off[nchunks];
    return true;
}
}  // namespace

bool WriteStdb(const Database &db, const std::string &path, bool packed, std::string *err) {
    // Solver.cs:454-462 ExportOutput: FileMode.Create, overwrite.  Entries are streamed in chunks, so
    // the writer is not bound by protobuf-net's 2 GB MemoryStream; the chunks of a library are encoded
    // and written in parallel, each at its own offset (write_lib): the bytes are SerializeStdb's.
    int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);   // (a write-only special file)
    if (fd < 0) { if (err) *err = "cannot open " + path + " for writing"; return false; }
    // Writing chunks at their offsets needs a file that can seek.  Anything else -- a FIFO, /dev/stdout into a pipe --
    // gets the bytes the way round 4 wrote them: one writer, in order (pwrite fails there with ESPIPE).
    struct stat sb;
    const bool seekable = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
    const int threads = seekable ? HostThreads() : 1;
    const Database::ResultView *rv = (db.results.disp || db.results.strain || db.results.fetch) ? &db.results : nullptr;
    // the small tail of the file first (its size is part of the bound)
    std::string tail;
    {
        W w{tail, packed};
        enc_lib_range(3, db.MatLib.Items(), 0, db.MatLib.Items().size(), packed, tail);
        enc_lib_range(4, db.BCLib.Items(), 0, db.BCLib.Items().size(), packed, tail);
        w.i32(5, db.nDOF);
        if (db.has_analysis) { std::string t; enc(db.AnalysisLib, packed, t); w.bytes(6, t); }
        if (db.has_info) { std::string t; enc(db.Info, packed, t); w.bytes(7, t); }
    }
    Sink sink;
    sink.fd = fd;
    sink.sequential = !seekable;
    // STAN_STDB_WRITE=map: through a shared mapping (see Sink) when the file system has room for the bound
    const char *mode = getenv("STAN_STDB_WRITE");
    if (seekable && mode && !strcmp(mode, "map")) {
        const size_t bound = max_lib_size(db.NodeLib.Items(), rv, threads) + max_lib_size(db.ElemLib.Items(), rv, threads) + tail.size() + 4096;
        struct statvfs vfs;
        const bool room = fstatvfs(fd, &vfs) == 0 && (double)vfs.f_bavail * (double)vfs.f_frsize > 1.05 * (double)bound;
        if (room && ftruncate(fd, (off_t)bound) == 0) {
            void *mp = mmap(nullptr, bound, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (mp != MAP_FAILED) { sink.map = (char *)mp; sink.map_len = bound; }
        }
    }
    int64_t off = 0;
    bool ok = write_lib(sink, &off, 1, db.NodeLib.Items(), packed, threads, rv);
    ok = ok && write_lib(sink, &off, 2, db.ElemLib.Items(), packed, threads, rv);
    ok = ok && sink.put(tail.data(), tail.size(), off);
    off += (int64_t)tail.size();
    if (sink.map) munmap(sink.map, sink.map_len);
    if (seekable && ftruncate(fd, ok ? (off_t)off : 0) != 0) ok = false;   // the true length (a failed export leaves an empty file, not a padded one)
    ok = (close(fd) == 0) && ok;
    if (!ok && err) *err = "short write to " + path;
    return ok;
}

namespace {
struct Span { const uint8_t *p, *end; };
// entries of one library, decoded on `threads` threads into wire order
template <typename T>
bool dec_lib_parallel(const std::vector<Span> &spans, OrderedDict<T> &lib, int threads, size_t *bad) {
    std::vector<std::pair<int, T>> items(spans.size());
    std::atomic<size_t> next{0};
    std::atomic<size_t> first_bad{(size_t)-1};
    constexpr size_t CHUNK = 2048;
    auto work = [&](int) {
        for (;;) {
            const size_t c = next.fetch_add(CHUNK);
            if (c >= spans.size()) return;
            const size_t e = std::min(spans.size(), c + CHUNK);
            for (size_t i = c; i < e; i++) {
                R r{spans[i].p, spans[i].end, true};
                if (!dec_entry(r, items[i].first, items[i].second)) {
                    size_t cur = first_bad.load();
                    while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
                }
            }
        }
    };
    if (threads <= 1 || spans.size() < 4 * CHUNK) work(0);
    else run_threads(threads, work);
    if (first_bad.load() != (size_t)-1) { *bad = first_bad.load(); return false; }
    lib.Adopt(std::move(items), threads);
    return true;
}
}  // namespace

bool ParseStdb(const uint8_t *data, size_t size, Database *db, std::string *err) {
    const bool trace = getenv("STAN_HOST_TRACE") != nullptr;   // stage times on stderr (diagnosis)
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t0 = now();
    auto lap = [&](const char *what) {
        if (trace) fprintf(stderr, "[stan host] ParseStdb %-28s %.3f s\n", what, std::chrono::duration<double>(now() - t0).count());
        t0 = now();
    };
    *db = Database();
    db->has_analysis = false;  // SkipConstructor: members absent from the wire stay null
    db->has_info = false;
    // pass 1: the top level only (tags and lengths); the payloads of the four libraries are remembered
    std::vector<Span> lib[4];
    R r{data, data + size, true};
    FIELDS(r)
        if (f >= 1 && f <= 4 && wt == 2) {
            R sub = r.sub();
            if (r.ok) lib[f - 1].push_back(Span{sub.p, sub.end});
        }
        else if (f == 5 && wt == 0) db->nDOF = (int)r.varint();
        else if (f == 6 && wt == 2) { db->has_analysis = true; if (!dec(r.sub(), db->AnalysisLib)) r.ok = false; }
        else if (f == 7 && wt == 2) { db->has_info = true; if (!dec(r.sub(), db->Info)) r.ok = false; }
        else r.skip(wt);
    END_FIELDS(r)
    const uint8_t *where = r.p;
    bool ok = r.ok;
    lap("top-level scan");
    // pass 2: the entries, in parallel (a repeated key keeps its first entry: OrderedDict::Adopt)
    const int threads = HostThreads();
    size_t bad = 0;
    if (ok && !dec_lib_parallel(lib[0], db->NodeLib, threads, &bad)) { ok = false; where = lib[0][bad].p; }
    lap("nodes: decode + index");
    if (ok && !dec_lib_parallel(lib[1], db->ElemLib, threads, &bad)) { ok = false; where = lib[1][bad].p; }
    lap("elements: decode + index");
    if (ok && !dec_lib_parallel(lib[2], db->MatLib, 1, &bad)) { ok = false; where = lib[2][bad].p; }
    if (ok && !dec_lib_parallel(lib[3], db->BCLib, 1, &bad)) { ok = false; where = lib[3][bad].p; }
    if (!ok && err) *err = "malformed STdb (protobuf wire error near byte " +
                           std::to_string((size_t)(where - data)) + ")";
    return ok;
}

bool ReadStdb(const std::string &path, Database *db, std::string *err) {
    // File.ReadAllBytes (Solver.cs:26).  Round 5: the file is mapped, not copied into a zero-filled buffer first (438 MB
    // at 148^3): the decoding threads take its pages straight from the page cache.
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) { if (err) *err = "cannot open " + path; return false; }
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); if (err) *err = "cannot read " + path; return false; }
    const size_t n = (size_t)sb.st_size;
    if (n == 0) { close(fd); return ParseStdb(nullptr, 0, db, err); }
    void *map = mmap(nullptr, n, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);   // (populate: no minor fault per page in the decoding threads)
    if (map == MAP_FAILED) {   // a file system without mmap: read it
        std::string buf(n, '\0');
        size_t got = 0;
        while (got < n) {
            const ssize_t k = read(fd, &buf[got], n - got);
            if (k < 0 && errno == EINTR) continue;
            if (k <= 0) break;
            got += (size_t)k;
        }
        close(fd);
        if (got != n) { if (err) *err = "cannot read " + path; return false; }
        return ParseStdb((const uint8_t *)buf.data(), buf.size(), db, err);
    }
    madvise(map, n, MADV_WILLNEED);
    const bool ok = ParseStdb((const uint8_t *)map, n, db, err);
    munmap(map, n);
    close(fd);
    return ok;
}

}  // namespace stan
