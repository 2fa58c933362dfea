"""The C# side of the drop-in (integration/*.cs) checked MECHANICALLY against include/stan_hip.h: no .NET toolchain
exists in this image, so nothing compiles those files; this test is what keeps them from drifting.
  * every function the header declares has exactly one [DllImport] and vice versa (nothing invented);
  * same arity, and every managed argument type is the blittable image of the C type;
  * the [StructLayout(Sequential)] records have the header's fields in the header's order and widths;
  * the constants the shim uses (error codes, element types, precision modes, options) equal the #defines;
  * the library really exports every one of those names."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "stan_hip.h")
SHIM = os.path.join(ROOT, "integration", "StanHip.cs")
SHIM2 = os.path.join(ROOT, "integration", "SolverFunctions.Hip.cs")


def _strip_c_comments(t):
    t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
    return re.sub(r"//[^\n]*", " ", t)


def _c_kind(decl):
    """C parameter declaration -> (kind, pointee): kind in scalar | ptr | handle | out_handle | struct."""
    d = re.sub(r"\bconst\b", " ", decl).strip()
    arr = re.search(r"\[\s*\d*\s*\]\s*$", d)
    if arr:
        d = d[:arr.start()].strip()
    stars = d.count("*")
    d = d.replace("*", " ")
    toks = d.split()
    base = toks[0] if toks[0] != "struct" else toks[1]
    nptr = stars + (1 if arr else 0)
    if base in ("stan_ctx", "stan_matrix", "stan_results", "void"):
        return ("out_handle" if nptr == 2 else "handle", base)
    if nptr == 2:      # `const double **p`: a raw pointer handed back (stan_hip_results_map) = out IntPtr
        return ("out_handle", base)
    if base in ("stan_matrix_info", "stan_profile"):
        assert nptr == 1
        return ("struct", base)
    norm = {"int": "int32", "int32_t": "int32", "int64_t": "int64", "double": "double", "float": "float",
            "uint8_t": "uint8", "char": "uint8"}[base]
    return ("scalar" if nptr == 0 else "ptr", norm)


def c_prototypes():
    t = _strip_c_comments(open(HEADER).read())
    t = t[t.index('extern "C"'):]
    t = re.sub(r"^\s*#.*$", " ", t, flags=re.M)                                   # preprocessor lines
    t = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", t, flags=re.S)   # the two records
    t = re.sub(r"typedef[^;]*;", " ", t)
    protos = {}
    for stmt in t.split(";"):
        m = re.search(r"((?:const\s+)?[A-Za-z_]\w*[\s\*]+)(stan_hip_\w+)\s*\((.*)\)\s*$", stmt.strip(), flags=re.S)
        if not m:
            continue
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3)
        ret_kind = {"int": "int32", "void": "void", "int64_t": "int64", "const char *": "handle"}[re.sub(r"\s+", " ", ret)]
        plist = [p.strip() for p in params.replace("\n", " ").split(",") if p.strip() and p.strip() != "void"]
        assert name not in protos, name
        protos[name] = (ret_kind, [_c_kind(p) for p in plist])
    return protos


CS_SCALAR = {"int": "int32", "long": "int64", "double": "double", "float": "float", "byte": "uint8"}


def _cs_kind(decl):
    d = re.sub(r"\[(Out|In|In, Out)\]", " ", decl).strip()
    toks = d.split()
    mod = toks[0] if toks[0] in ("out", "ref") else None
    typ = toks[1] if mod else toks[0]
    if typ == "IntPtr":
        return ("out_handle" if mod else "handle", None)
    if typ in ("StanMatrixInfo", "StanProfile"):
        assert mod in ("out", "ref")
        return ("struct", {"StanMatrixInfo": "stan_matrix_info", "StanProfile": "stan_profile"}[typ])
    if typ.endswith("[]"):
        return ("ptr", CS_SCALAR[typ[:-2]])
    return (("ptr" if mod else "scalar"), CS_SCALAR[typ])


def cs_imports():
    t = _strip_c_comments(open(SHIM).read())
    out = {}
    for m in re.finditer(r"\[DllImport\(Lib\)\]\s*internal\s+static\s+extern\s+(\w+)\s+(stan_hip_\w+)\s*\(([^;]*?)\)\s*;", t, flags=re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        assert name not in out, "declared twice: " + name
        ret_kind = {"int": "int32", "void": "void", "long": "int64", "IntPtr": "handle"}[ret]
        plist = [p.strip() for p in params.replace("\n", " ").split(",") if p.strip()]
        out[name] = (ret_kind, [_cs_kind(p) for p in plist])
    return out


def test_every_header_function_has_one_matching_dllimport():
    c, cs = c_prototypes(), cs_imports()
    assert len(c) >= 29, sorted(c)
    assert sorted(c) == sorted(cs), (sorted(set(c) - set(cs)), sorted(set(cs) - set(c)))
    for name, (ret, params) in c.items():
        cret, cparams = cs[name]
        assert ret == cret, (name, ret, cret)
        assert len(params) == len(cparams), (name, len(params), len(cparams))
        for i, ((k, b), (ck, cb)) in enumerate(zip(params, cparams)):
            if ck == "handle" and k in ("handle", "ptr"):     # IntPtr: an opaque handle or a device / raw pointer
                continue
            assert (k, b) == (ck, cb) or (k == ck == "out_handle"), (name, i, (k, b), (ck, cb))


def _c_struct(name):
    t = _strip_c_comments(open(HEADER).read())
    body = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), t, flags=re.S).group(1)
    fields = []
    for stmt in body.split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        typ, names = stmt.split(None, 1)
        for n in names.split(","):
            fields.append((n.strip(), {"int64_t": "int64", "int32_t": "int32", "double": "double", "float": "float"}[typ]))
    return fields


def _cs_struct(name):
    t = _strip_c_comments(open(SHIM).read())
    m = re.search(r"\[StructLayout\(LayoutKind\.Sequential\)\]\s*public\s+struct\s+%s\s*\{(.*?)\}" % name, t, flags=re.S)
    return [(n, CS_SCALAR[ty]) for ty, n in re.findall(r"public\s+(\w+)\s+(\w+)\s*;", m.group(1))]


def test_struct_records_match_the_header_field_by_field():
    assert _cs_struct("StanMatrixInfo") == _c_struct("stan_matrix_info")
    assert _cs_struct("StanProfile") == _c_struct("stan_profile")
    # ... and the ctypes mirror the Python tests use
    from stan_amd import hip
    cmap = {ctypes.c_int64: "int64", ctypes.c_int32: "int32", ctypes.c_double: "double", ctypes.c_float: "float"}
    assert [(n, cmap[t]) for n, t in hip.MatrixInfo._fields_] == _c_struct("stan_matrix_info")
    assert [(n, cmap[t]) for n, t in hip.Profile._fields_] == _c_struct("stan_profile")


def test_constants_of_the_shim_are_the_header_defines():
    h = open(HEADER).read()
    defs = {m.group(1): int(m.group(2).strip("()")) for m in re.finditer(r"#define\s+(STAN_\w+)\s+(\(?-?\d+\)?)", h)}
    t = _strip_c_comments(open(SHIM).read())
    used = {m.group(1): int(m.group(2)) for m in re.finditer(r"\b(STAN_[A-Z0-9_]+)\s*=\s*(-?\d+)", t)}
    assert len(used) >= 30
    for k, v in used.items():
        assert defs.get(k) == v, (k, v, defs.get(k))
    # every option the header defines is available to the managed side
    assert {k for k in defs if k.startswith("STAN_OPT_")} <= set(used)


def test_the_library_exports_every_imported_name(built_libs):
    lib = ctypes.CDLL(os.path.join(ROOT, "stan_amd", "lib", "libstan_hip.so"))
    for name in cs_imports():
        assert hasattr(lib, name), name


def test_replacement_methods_call_only_declared_imports_with_declared_arity():
    """integration/SolverFunctions.Hip.cs: every StanHipNative.<f>(...) call names a declared import and passes as many
    arguments as it takes (the write-back of the 8x6 blocks into dE / dS is there, in front of Update_StrainStress)."""
    cs = cs_imports()
    t = _strip_c_comments(open(SHIM2).read())
    calls = 0
    for m in re.finditer(r"StanHipNative\.(stan_hip_\w+)\s*\(", t):
        name = m.group(1)
        assert name in cs, name
        depth, i, args, cur = 1, m.end(), [], ""
        while depth:
            ch = t[i]
            if ch in "([":
                depth += 1
            elif ch in ")]":
                depth -= 1
                if depth == 0:
                    break
            if ch == "," and depth == 1:
                args.append(cur); cur = ""
            else:
                cur += ch
            i += 1
        if cur.strip():
            args.append(cur)
        assert len(args) == len(cs[name][1]), (name, len(args), len(cs[name][1]))
        calls += 1
    assert calls >= 6
    assert re.search(r"e\.dE\[a\]\.SetFast\(c, 0, strain\[48 \* i \+ 6 \* a \+ c\]\)", t)
    assert re.search(r"e\.dS\[a\]\.SetFast\(c, 0, stress\[48 \* i \+ 6 \* a \+ c\]\)", t)


REF = "/root/reference/src"


def test_shim_uses_only_members_the_reference_declares():
    """integration/SolverFunctions.Hip.cs is written against the reference's object model without a compiler to check
    it: every member it touches must exist, public, in the reference's sources (checked HERE, where the checkout is
    present; the GPU box has none and skips).  Reads the reference, copies nothing."""
    import pytest
    if not os.path.isdir(REF):
        pytest.skip("no reference checkout on this machine")

    def src(rel):
        return open(os.path.join(REF, rel), encoding="utf-8-sig").read()
    node, elem, mat = src("STAN_Database/Node.cs"), src("STAN_Database/Element.cs"), src("STAN_Database/Material.cs")
    db, ana, mst = src("STAN_Database/Database.cs"), src("STAN_Database/Analysis.cs"), src("STAN_Database/MatrixST.cs")
    need = [
        (node, r"public\s+int\s+ID\b"), (node, r"public\s+double\s+X\b"), (node, r"public\s+double\s+Y\b"),
        (node, r"public\s+double\s+Z\b"), (node, r"public\s+int\[\]\s+DOF\b"), (node, r"public\s+double\[\]\s+dU_buffer\b"),
        (elem, r"public\s+int\s+ID\b"), (elem, r"public\s+string\s+Type\b"), (elem, r"public\s+int\s+MatID\b"),
        (elem, r"public\s+List<int>\s+NList\b"), (elem, r"public\s+MatrixST\[\]\s+dE\b"), (elem, r"public\s+MatrixST\[\]\s+dS\b"),
        (elem, r"public\s+void\s+Update_StrainStress\(int\s+\w+\)"), (elem, r"public\s+void\s+Initialize_Increment\(int\s+\w+\)"),
        (mat, r"public\s+int\s+ID\b"), (mat, r"public\s+double\s+E\b"), (mat, r"public\s+double\s+Poisson\b"),
        (db, r"public\s+Dictionary<int,\s*Node>\s+NodeLib\b"), (db, r"public\s+Dictionary<int,\s*Element>\s+ElemLib\b"),
        (db, r"public\s+Dictionary<int,\s*Material>\s+MatLib\b"), (db, r"public\s+int\s+nDOF\b"),
        (db, r"public\s+Analysis\s+AnalysisLib\b"),
        (ana, r"public\s+double\s+GetLinSolverTolerance\(\)"), (ana, r"public\s+int\s+GetLinSolverMaxIter\(\)"),
        (ana, r"public\s+string\s+GetLinSolver\(\)"),
        (mst, r"public\s+void\s+SetFast\(int\s+\w+,\s*int\s+\w+,\s*double\s+\w+\)"),
    ]
    for text, pat in need:
        assert re.search(pat, text), pat
    # the call sites the shim replaces are where INTEGRATION.md says they are
    solver = src("STAN_Solver/Solver.cs").split("\n")
    assert "Fun.ParallelAssembly_K(DB, nDOF_reduction, inc" in solver[155]            # Solver.cs:156
    assert "Fun.LinearSolver_CG(K, F, DB.AnalysisLib)" in solver[161]                 # Solver.cs:162
    assert "Elem.Recovery_Stress(DB)" in solver[185] and "E.Update_StrainStress(inc)" in solver[208]
    fun = src("STAN_Solver/SolverFunctions.cs").split("\n")
    assert "ParallelAssembly_K(" in fun[116] and "LinearSolver_CG(" in fun[269]       # SolverFunctions.cs:117, :270
    # and every member access of the shim on those objects is in the list above
    shim = _strip_c_comments(open(SHIM2).read())
    for m in set(re.findall(r"\b(?:n|e|m)\.(\w+)", shim)):
        assert m in {"ID", "X", "Y", "Z", "DOF", "dU_buffer", "Type", "MatID", "NList", "dE", "dS", "E", "Poisson"}, m
    for m in set(re.findall(r"\bDB\.(\w+)", shim)):
        assert m in {"NodeLib", "ElemLib", "MatLib", "nDOF"}, m
    for m in set(re.findall(r"\bAnalysisLib\.(\w+)", shim)):
        assert m in {"GetLinSolverTolerance", "GetLinSolverMaxIter"}, m


def _method_body(text, name):
    m = re.search(r"public\s+[\w\.\[\]<>]+\s+%s\s*\(" % name, text)
    i = text.index("{", m.end())
    depth, j = 1, i + 1
    while depth:
        depth += {"{": 1, "}": -1}.get(text[j], 0)
        j += 1
    return text[i:j]


def _console_literals(body):
    """The string literals of every Console.Write / Console.WriteLine call of a method body, in order."""
    out = []
    for m in re.finditer(r"Console\.Write(?:Line)?\s*\(", body):
        depth, i = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(body[i], 0)
            i += 1
        out += re.findall(r'"((?:[^"\\]|\\.)*)"', body[m.end():i - 1])
    return out


# what the two hot methods of the reference print, literal by literal (SolverFunctions.cs:127, 177; :273, 323-327)
CONSOLE_REF = {
    "ParallelAssembly_K": ["   K Matrix assembly: ", "          Done in ", "F2", "s"],
    "LinearSolver_CG": ["   Solving linear system...   ", "  NORMAL ", "  ERROR ", " (type ", ")", " in ", "F2", "s"],
}


def test_the_shim_prints_the_reference_console_lines_literal_by_literal():
    """`integration/SolverFunctions.Hip.cs` printed " (type N, M iterations)" where the reference
    prints " (type N)" (SolverFunctions.cs:325).  Every Console.Write* literal of the two replaced methods is compared
    with the reference's -- against the list above everywhere, and against the reference's own source where the
    checkout is present (this container; the GPU box has none)."""
    shim = _strip_c_comments(open(SHIM2).read())
    for name, want in CONSOLE_REF.items():
        assert _console_literals(_method_body(shim, name)) == want, name
    if os.path.isdir(REF):
        ref = _strip_c_comments(open(os.path.join(REF, "STAN_Solver", "SolverFunctions.cs"), encoding="utf-8-sig").read())
        for name, want in CONSOLE_REF.items():
            assert _console_literals(_method_body(ref, name)) == want, name
    # the native console driver (stan_amd/host/solver_functions.cpp) prints the same text through printf
    cpp = open(os.path.join(ROOT, "stan_amd", "host", "solver_functions.cpp")).read()
    for lit in ('"   K Matrix assembly: "', '"          Done in %.2fs\\n"', '"   Solving linear system...   "',
                '"  NORMAL "', '"  ERROR "', '" (type %d) in %.2fs\\n"'):
        assert lit in cpp, lit
