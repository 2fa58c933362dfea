// bdf.cpp -- Nastran short-format mesh import, with the reference's quirks kept:
// Database.ReadNastranMesh (Database.cs:39-111), Node(string) (Node.cs:25-80),
// Element(string) (Element.cs:35-73).  SURVEY.md section 8(f) rank 3.
#include <cerrno>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "model.h"

namespace stan {
namespace {

bool contains(const std::string &s, const char *t) { return s.find(t) != std::string::npos; }
std::string replace_all(std::string s, const std::string &a, const std::string &b) {
    size_t p = 0;
    while ((p = s.find(a, p)) != std::string::npos) { s.replace(p, a.size(), b); p += b.size(); }
    return s;
}
// int.Parse / int.TryParse: optional whitespace, sign, decimal digits, nothing else
bool parse_int(const std::string &s, int *out) {
    if (s.empty()) return false;
    errno = 0;
    char *end = nullptr;
    const long v = strtol(s.c_str(), &end, 10);
    if (end == s.c_str() || errno || v < INT32_MIN || v > INT32_MAX) return false;
    while (*end == ' ' || *end == '\t') end++;
    if (*end) return false;
    for (char c : s) if (c == '.' || c == 'e' || c == 'E') return false;
    *out = (int)v;
    return true;
}
bool parse_double(const std::string &s, double *out) {  // double.Parse(InvariantCulture)
    if (s.empty()) return false;
    if (s.find_first_of("xXpP") != std::string::npos) return false;  // no hex floats in .NET
    char *end = nullptr;
    const double v = strtod(s.c_str(), &end);
    if (end == s.c_str() || *end) return false;
    *out = v;
    return true;
}

std::vector<std::string> read_all_lines(const std::string &path, bool *ok) {  // File.ReadAllLines
    std::ifstream in(path, std::ios::binary);
    std::vector<std::string> lines;
    *ok = (bool)in;
    if (!in) return lines;
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string all = ss.str();
    std::string cur;
    for (size_t i = 0; i < all.size(); i++) {
        const char c = all[i];
        if (c == '\n' || c == '\r') {
            lines.push_back(cur);
            cur.clear();
            if (c == '\r' && i + 1 < all.size() && all[i + 1] == '\n') i++;
        } else
            cur.push_back(c);
    }
    if (!cur.empty()) lines.push_back(cur);
    return lines;
}

}  // namespace

// Node.cs:25-80
bool Node::FromBdfLine(const std::string &input, Node *out) {
    std::vector<std::string> data;
    for (size_t i = 0; i < input.size() / 8; i++) {  // integer division drops a ragged tail
        std::string text = replace_all(input.substr(i * 8, 8), " ", "");
        bool blank = true;
        for (char c : text) if (c != ' ' && c != '\t') blank = false;
        if (blank) continue;  // blank fields (e.g. CP) are skipped, not kept as placeholders
        if (!contains(text, "e") && !contains(text, "E")) {
            if (contains(text.substr(1), "-")) {  // "7.11-15" -> "7.11e-15"
                if (text[0] == '-') text = "-" + replace_all(text.substr(1), "-", "e-");
                else text = replace_all(text, "-", "e-");
            }
            // Node.cs:52-55: text.Replace("+", "e+") -- the result is discarded in the
            // reference, so "1.5+3" stays unparseable; kept.
        }
        if (text[0] == '.') text = "0" + text;
        data.push_back(text);
    }
    if (data.size() < 5) return false;  // data[4] would throw ArgumentOutOfRange
    Node n;
    if (!parse_int(data[1], &n.ID)) return false;
    if (!parse_double(data[2], &n.X) || !parse_double(data[3], &n.Y) || !parse_double(data[4], &n.Z))
        return false;
    n.DOF = {0, 0, 0};  // new int[3]
    n.DispX = {0.0}; n.DispY = {0.0}; n.DispZ = {0.0};
    *out = n;
    return true;
}

// Element.cs:35-73
bool Element::FromBdfLine(const std::string &input, Element *out) {
    // Regex.Split(input, @"\s+"): leading whitespace yields an empty first token
    std::vector<std::string> data;
    {
        std::string cur;
        bool in_ws = false;
        for (char c : input) {
            const bool ws = c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\f' || c == '\v';
            if (ws) { if (!in_ws) { data.push_back(cur); cur.clear(); } in_ws = true; }
            else { cur.push_back(c); in_ws = false; }
        }
        data.push_back(cur);
    }
    if (data.size() < 3) return false;
    Element e;
    if (!parse_int(data[1], &e.ID) || !parse_int(data[2], &e.PID)) return false;
    for (size_t i = 3; i < data.size(); i++) {
        const std::string t = replace_all(data[i], "+", "");
        int v;
        if (parse_int(t, &v)) e.NList.push_back(v);
    }
    if (data[0] == "CHEXA") { e.Type = "HEX8_G2"; e.has_type = true; }
    if (data[0] == "CPENTA") { e.Type = "PENTA6_G2"; e.has_type = true; }
    if (data[0] == "CTETRA") { e.Type = "TET4_G2"; e.has_type = true; }
    e.MatID = 0;
    *out = e;
    return true;
}

// Database.cs:39-111 (Part creation is GUI-side and not part of the STdb)
bool Database::ReadNastranMesh(const std::string &path, std::string *err) {
    bool ok;
    std::vector<std::string> data = read_all_lines(path, &ok);
    if (!ok) { if (err) *err = "cannot open " + path; return false; }
    conn_index.clear();   // (cache of AssignDOF: the mesh changes)
    for (size_t i = 0; i < data.size(); i++) {
        if (data[i].rfind("$", 0) == 0) continue;  // commented line
        if (contains(data[i], "CHEXA")) {           // Elem_types_allowed = { "CHEXA" }
            std::string temp = data[i];
            for (size_t j = i + 1; j < data.size(); j++) {
                if (data[j].rfind("+", 0) == 0 || data[j].rfind(" ", 0) == 0) { temp += data[j]; i = j; }
                else break;
            }
            Element E;
            if (!Element::FromBdfLine(temp, &E) || !ElemLib.Add(E.ID, E)) Import_Error.push_back(temp);
        }
        if (data[i].rfind("GRID", 0) == 0) {
            Node N;
            if (!Node::FromBdfLine(data[i], &N) || !NodeLib.Add(N.ID, N)) Import_Error.push_back(data[i]);
        }
    }
    return true;
}

}  // namespace stan
