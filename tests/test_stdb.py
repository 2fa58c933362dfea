"""CPU tests of the STAN_Database mirror: STdb codec, bdf import, and the plumbing config
(BASELINE.json configs[0]): .bdf -> Database -> STdb -> solve (oracle, CPU) -> STdb."""
import os
import struct

import numpy as np
import pytest

from stan_amd import bdf, host, problem
from stan_amd.cube import cube_bcs, cube_mesh


def _cube_db(n=3, jitter=0.0, tmp=None, etype="HEX8_G2", tol=1e-12):
    xyz, conn = cube_mesh(n, jitter=jitter)
    d = host.Db()
    if tmp is not None:
        path = os.path.join(str(tmp), "mesh.bdf")
        bdf.write_bdf(path, xyz, conn)
        assert d.read_bdf(path) == 0
    else:
        ne = conn.shape[0]
        d.set_mesh(np.arange(1, xyz.shape[0] + 1), xyz, np.arange(1, ne + 1), np.ones(ne), conn + 1, etype)
    d.add_material(1, "Steel", 210000.0, 0.3)
    d.assign_part(1, 1, etype)
    spc, ld, f = cube_bcs(n, clamp_faces="xyz" if etype == "HEX8_G1" else "x")
    d.add_bc(1, "fix", "SPC", spc + 1, np.ones((len(spc), 3)))
    d.add_bc(2, "load", "PointLoad", ld + 1, np.tile(f, (len(ld), 1)))
    d.set_analysis(tol=tol)
    return d, xyz, conn


def test_roundtrip_unpacked_and_packed(built_libs):
    d, _, _ = _cube_db(3, jitter=0.1)
    b = d.serialize()
    d2 = host.Db.parse_stdb(b)
    assert d2.serialize() == b                      # encode -> decode -> encode byte-identical
    bp = d.serialize(packed=True)
    assert len(bp) < len(b)
    d3 = host.Db.parse_stdb(bp)                      # reader accepts packed
    assert d3.serialize() == b and d3.serialize(packed=True) == bp
    assert d2.sizes() == d.sizes() and d2.analysis() == d.analysis()


def test_wire_details(built_libs):
    d = host.Db()
    d.set_mesh([5, 7], [[0.0, 0, 0], [1.5, -2.0, 0]], [], [], np.zeros((0, 8)))
    d.has = None
    b = d.serialize()
    # first NodeLib entry: field 1 LEN { key=5 (08 05), value { ID=5, DOF=[0,0,0], Disp=[0] } }
    # X=Y=Z=0 are omitted (implicit zero default), list elements are written even when zero
    assert b[0] == 0x0A
    entry = b[2:2 + b[1]]
    assert entry[:2] == b"\x08\x05" and entry[2] == 0x12
    val = entry[4:4 + entry[3]]
    assert val[:2] == b"\x08\x05"
    assert b"\x11" not in val[2:3]                 # no field 2 (X) right after ID
    assert val.count(b"\x30\x00") == 3             # DOF = {0,0,0} unpacked
    assert struct.pack("<d", 1.5) in b and struct.pack("<d", -2.0) in b
    # root nDOF = 6 -> field 5 varint
    assert b"\x28\x06" in b


def test_reader_tolerates_unknown_fields_and_swapped_map_entries(built_libs):
    d, _, _ = _cube_db(1)
    b = d.serialize()
    # append an unknown varint field 15 and an unknown length-delimited field 14 at root level
    b2 = b + b"\x78\x2a" + b"\x72\x03abc"
    assert host.Db.parse_stdb(b2).serialize() == b
    # a map entry with value before key: MatLib {2: value, 1: key}
    import re
    m = host.Db.parse_stdb(b)
    raw = bytearray()
    mat = b"\x08\x01\x12\x07Elastic\x1a\x05Steel\x21" + struct.pack("<d", 210000.0) + b"\x29" + struct.pack("<d", 0.3) + b"\x30\x01"
    entry = b"\x12" + bytes([len(mat)]) + mat + b"\x08\x01"
    raw += b"\x1a" + bytes([len(entry)]) + entry
    mm = host.Db.parse_stdb(bytes(raw))
    assert mm.sizes()["materials"] == 1
    # truncated input is an error, not a crash
    with pytest.raises(host.StanHostError) as ei:
        host.Db.parse_stdb(b[:len(b) // 2 + 1])
    assert ei.value.code == -23


def test_negative_int_is_ten_byte_varint(built_libs):
    d = host.Db()
    d.set_mesh([-3], [[0.0, 0, 0]], [], [], np.zeros((0, 8)))
    b = d.serialize()
    assert b"\x08\xfd\xff\xff\xff\xff\xff\xff\xff\xff\x01" in b
    assert host.Db.parse_stdb(b).flat()["node_ids"].tolist() == [-3]


def test_bdf_import_quirks(built_libs, tmp_path):
    p = tmp_path / "q.bdf"
    p.write_text(
        "$$  GRID Data\n"
        "GRID           1             0.0    15.0     0.0\n"
        "GRID           2        -7.11-15     5.0     0.0\n"      # README.md:37 shorthand exponent
        "GRID    %8d%8s%8s%8s%8s\n" % (3, "", ".5", "-.25", "1.5e1") +  # leading '.', explicit e
        "GRID           4             1.0     2.0     3.0 RAGGED\n"  # Length/8 drops the ragged tail
        "GRID           5             1.0     2.0\n"              # too few fields -> import error
        "GRID           1             9.0     9.0     9.0\n"      # duplicate ID -> import error
        "$ GRID        99             1.0     1.0     1.0\n"      # comment
        "CHEXA          1       1       1       2       3       4       1       2+       \n"
        "+              3       4\n"
        "CHEXA          2       7       4       3       2       1       4       3\n"
        "               2       1\n"                              # continuation starting with a blank
        "CTETRA         3       1       1       2       3       4\n"   # not admitted (Database.cs:44-48)
    )
    d = host.Db()
    nerr = d.read_bdf(str(p))
    s = d.sizes()
    assert (s["nodes"], s["elements"], nerr, s["nDOF"]) == (4, 2, 2, 12)
    f = host.Db.parse_stdb(d.serialize())
    # flatten needs materials; check raw coordinates through the wire instead
    b = d.serialize()
    assert struct.pack("<d", -7.11e-15) in b and struct.pack("<d", 0.5) in b
    assert struct.pack("<d", -0.25) in b and struct.pack("<d", 15.0) in b
    assert b"HEX8_G2" in b and b"TET4" not in b
    assert f.sizes() == s | {"import_errors": 0}


def test_db_matches_flat_problem_setup(built_libs, tmp_path):
    """Database path (bdf -> AssignDOF -> BC tables) == the array path bench.py uses."""
    n = 4
    d, xyz, conn = _cube_db(n, tmp=tmp_path)
    d.assign_dof()
    fl = d.flat()
    job = problem.cube_job(n)
    assert np.array_equal(fl["conn"], job.conn) and np.allclose(fl["xyz"], job.xyz)
    assert np.array_equal(fl["node_dof"], job.node_dof)       # bit-exact DOF numbering
    red, nfix, F = d.reduction()
    assert nfix == job.n_fixed and np.array_equal(red, job.red) and np.array_equal(F, job.F)
    assert fl["elem_type"].tolist() == [2] * n ** 3 and np.allclose(fl["mat_E_nu"], [[210000.0, 0.3]])


def test_flatten_errors(built_libs):
    d, _, _ = _cube_db(1)
    with pytest.raises(host.StanHostError):      # DOF not assigned yet is fine, but MatID 0:
        d2 = host.Db()
        d2.set_mesh([1, 2, 3, 4, 5, 6, 7, 8], cube_mesh(1)[0], [1], [1], cube_mesh(1)[1] + 1)
        d2.assign_dof()
        d2.flat()                                  # MatLib[0] -> KeyNotFound in the reference
    d3 = host.Db()
    d3.set_mesh([1, 2, 3, 4, 5, 6, 7, 8], cube_mesh(1)[0], [1], [1], cube_mesh(1)[1] + 1, "TET4_G2")
    d3.add_material(1, "m", 1.0, 0.3)
    d3.assign_part(1, 1, "TET4_G2")
    d3.assign_dof()
    with pytest.raises(host.StanHostError):
        d3.flat()                                  # unsupported element Type


def test_plumbing_bdf_to_stdb_to_results(built_libs, oracle, tmp_path):
    """configs[0]: bdf -> STdb -> CPU reference algorithm (the oracle) -> STdb, no GPU."""
    n = 3
    d, xyz, conn = _cube_db(n, jitter=0.0, tmp=tmp_path)
    path = str(tmp_path / "model.STdb")
    d.write_stdb(path)
    m = host.Db.read_stdb(path)                       # what Solver.Main does
    m.assign_dof()
    fl = m.flat()
    red, nfix, F = m.reduction()
    rc, A = oracle.assemble(fl["xyz"], fl["node_dof"], fl["conn"], fl["elem_mat"], fl["elem_type"],
                            fl["mat_E_nu"], red)
    assert rc == 0
    U, rep = oracle.cg(A, F, m.analysis()["tol"])
    disp = host.nodal_displacements(fl["node_dof"], red, U)
    strain = np.zeros((n ** 3, 8, 6)); stress = np.zeros((n ** 3, 8, 6))
    for e in range(n ** 3):
        rc, strain[e], stress[e] = oracle.recover_hex8(fl["xyz"][fl["conn"][e]], 210000.0, 0.3, 2,
                                                       disp[fl["conn"][e]].ravel())
        assert rc == 0
    m.set_results(disp, strain, stress)
    m.write_stdb(path)                                # ExportOutput overwrites the input
    r = host.Db.read_stdb(path)
    assert r.sizes()["result_step"] == 1
    d1, e1, s1 = r.results(1)
    assert np.array_equal(d1, disp) and np.array_equal(e1, strain) and np.array_equal(s1, stress)
    d0, e0, s0 = r.results(0)
    assert not d0.any() and not e0.any() and not s0.any()
    # SURVEY.md Appendix D-like sanity: free end moves in +z under the (0,0,50) loads
    assert disp[:, 2].max() > 0 and abs(disp[cube_bcs(n)[0]]).max() == 0
    # the EList the reference serializes after AssignDOF is present too
    assert r.serialize() == m.serialize()


def test_results_file_streams_through_the_flush_boundary(built_libs, tmp_path):
    """the result payload is ~1.7 KB per element (2 x 2 MatrixST of
    8x6, unpacked doubles); the writer streams entries instead of building one buffer
    (protobuf-net's MemoryStream caps the reference at 2 GB).  24^3 = 13 824 elements ->
    ~27 MB, several 1 MiB flushes; file == in-memory serialisation, and it reads back."""
    import time
    n = 24
    d, xyz, conn = _cube_db(n)
    d.assign_dof()
    rng = np.random.default_rng(0)
    disp = rng.standard_normal((xyz.shape[0], 3))
    strain = rng.standard_normal((n ** 3, 8, 6))
    stress = rng.standard_normal((n ** 3, 8, 6))
    d.set_results(disp, strain, stress)
    path = str(tmp_path / "big.STdb")
    t0 = time.perf_counter()
    d.write_stdb(path)
    t1 = time.perf_counter()
    size = os.path.getsize(path)
    assert size > 20e6
    assert open(path, "rb").read() == d.serialize()
    r = host.Db.read_stdb(path)
    t2 = time.perf_counter()
    d1, e1, s1 = r.results(1)
    assert np.array_equal(d1, disp) and np.array_equal(e1, strain) and np.array_equal(s1, stress)
    assert (t1 - t0) < 60 and (t2 - t1) < 60      # (first-touch page faults dominate in a small VM; seconds elsewhere)
    # packed encoding is ~10 % smaller and reads back identically
    d.write_stdb(path, packed=True)
    assert os.path.getsize(path) < 0.95 * size
    assert host.Db.read_stdb(path).serialize() == d.serialize()


def test_flat_result_writer_writes_the_object_path_bytes(built_libs, tmp_path):
    """stan_host_db_write_stdb_with_results (what stan_solver uses by default): the results go from the flat
    arrays straight into the encoder.  Same file as initialising / updating every Node and Element object
    first (Solver.cs:81-90, 203-210) and serialising those, unpacked and packed, including negative zeros."""
    n = 9
    d, xyz, conn = _cube_db(n)
    d.assign_dof()
    rng = np.random.default_rng(3)
    disp = rng.standard_normal((xyz.shape[0], 3))
    disp[0, 0] = -0.0; disp[1, 1] = 0.0
    strain = rng.standard_normal((n ** 3, 8, 6))
    stress = rng.standard_normal((n ** 3, 8, 6))
    stress[0, 0, 0] = -0.0
    for packed in (False, True):
        a, b = str(tmp_path / "flat.STdb"), str(tmp_path / "obj.STdb")
        d.write_stdb_with_results(a, disp, strain, stress, packed=packed)
        before = d.serialize()
        d2, _, _ = _cube_db(n)
        d2.assign_dof()
        d2.set_results(disp, strain, stress)
        d2.write_stdb(b, packed=packed)
        assert open(a, "rb").read() == open(b, "rb").read()
        assert d.serialize() == before            # the database itself is unchanged
    r = host.Db.read_stdb(a)
    d1, e1, s1 = r.results(1)
    assert np.array_equal(d1, disp + 0.0) and np.array_equal(e1, strain) and np.array_equal(s1, stress)


def test_mapped_export_writes_the_bytes_of_the_write_calls(built_libs, tmp_path):
    """Round 5: large result files are written through a shared mapping of an upper-bound length that is cut to the true
    length at the end (buffered write calls to ONE file serialise on its inode lock whatever the thread count:
    stdb.cpp Sink).  STAN_STDB_WRITE forces either path on a small model: same bytes, packed and unpacked, with and
    without results."""
    import subprocess
    import sys
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
from stan_amd import host
from stan_amd.cube import cube_mesh, cube_bcs
n = 9
xyz, conn = cube_mesh(n)
d = host.Db()
ne = conn.shape[0]
d.set_mesh(np.arange(1, xyz.shape[0] + 1), xyz, np.arange(1, ne + 1), np.ones(ne), conn + 1, "HEX8_G2")
d.add_material(1, "Steel", 210000.0, 0.3); d.assign_part(1, 1, "HEX8_G2")
spc, ld, f = cube_bcs(n)
d.add_bc(1, "fix", "SPC", spc + 1, np.ones((len(spc), 3)))
d.add_bc(2, "load", "PointLoad", ld + 1, np.tile(f, (len(ld), 1)))
d.set_analysis(tol=1e-6)
rng = np.random.default_rng(5)
disp, strain, stress = rng.standard_normal((xyz.shape[0], 3)), rng.standard_normal((ne, 8, 6)), rng.standard_normal((ne, 8, 6))
for packed in (False, True):
    d.write_stdb(sys.argv[1] + "_plain_%%d" %% packed, packed=packed)
    d.write_stdb_with_results(sys.argv[1] + "_res_%%d" %% packed, disp, strain, stress, packed=packed)
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for mode in ("pwrite", "map"):
        base = str(tmp_path / mode)
        p = subprocess.run([sys.executable, "-c", code, base], env=dict(os.environ, STAN_STDB_WRITE=mode, STAN_HOST_THREADS="4"),
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        out[mode] = {k: open(base + k, "rb").read() for k in ("_plain_0", "_plain_1", "_res_0", "_res_1")}
    for k in out["map"]:
        assert len(out["map"][k]) > 10000 and out["map"][k] == out["pwrite"][k], k


def test_repeated_key_keeps_its_first_entry_also_on_the_parallel_index_build(built_libs, tmp_path):
    """Dictionary semantics of the libraries (OrderedDict::Adopt): a key that appears twice on the wire keeps its FIRST
    entry, position and all.  Round 5 builds the key index of a large library on the host threads (compare-and-swap
    insertion, the smaller position wins); 42^3 = 79 507 nodes is above the threshold of that path."""
    import struct
    n = 42
    xyz, conn = cube_mesh(n)
    d = host.Db()
    ne = conn.shape[0]
    d.set_mesh(np.arange(1, xyz.shape[0] + 1), xyz, np.arange(1, ne + 1), np.ones(ne), conn + 1, "HEX8_G2")
    d.add_material(1, "Steel", 210000.0, 0.3); d.assign_part(1, 1, "HEX8_G2")
    d.set_analysis(tol=1e-6)
    path = str(tmp_path / "m.STdb")
    d.write_stdb(path)
    base = open(path, "rb").read()

    def node_entry(key, x):       # Database field 1: map entry {1: key, 2: Node{1: ID, 2: X}}
        node = b"\x08" + bytes([key]) + b"\x11" + struct.pack("<d", x) + b"\x30\x00\x30\x01\x30\x02"    # + DOF = [0, 1, 2]
        entry = b"\x08" + bytes([key]) + b"\x12" + bytes([len(node)]) + node
        return b"\x0a" + bytes([len(entry)]) + entry
    os.environ["STAN_HOST_THREADS"] = "4"
    try:
        late = host.Db.parse_stdb(base + node_entry(5, 777.0))      # the repeat comes last: dropped
        early = host.Db.parse_stdb(node_entry(5, 777.0) + base)     # the repeat comes first: it is the entry
    finally:
        del os.environ["STAN_HOST_THREADS"]
    ref = host.Db.parse_stdb(base)
    nn = xyz.shape[0]
    assert ref.sizes()["nodes"] == late.sizes()["nodes"] == early.sizes()["nodes"] == nn
    fr, fl, fe = ref.flat(), late.flat(), early.flat()
    assert np.array_equal(fl["node_ids"], fr["node_ids"]) and np.array_equal(fl["xyz"], fr["xyz"])
    assert fe["node_ids"][0] == 5 and fe["xyz"][0, 0] == 777.0 and fe["xyz"][0, 1] == 0.0
    assert np.array_equal(np.delete(fe["node_ids"], 0), np.delete(fr["node_ids"], 4))
    # lookups go through the index either way: AssignDOF resolves 8 node IDs per element
    late.assign_dof()
    ref.assign_dof()
    assert np.array_equal(late.flat()["node_dof"], ref.flat()["node_dof"])


def test_export_into_a_pipe_writes_the_same_bytes(built_libs, tmp_path):
    """the export writes its chunks at their offsets (pwrite) and cuts the file to length at the end, which a
    target that cannot seek -- a FIFO, /dev/stdout into a pipe -- refuses (ESPIPE: "short write").  Such a target gets the
    bytes from one writer, in order: a model with results (many chunks, four host threads) written into a FIFO arrives
    byte for byte as the regular file does."""
    import threading
    import numpy as np
    from stan_amd import host
    from stan_amd.cube import cube_bcs, cube_mesh
    n = 18          # 5832 elements: more than one chunk of 4096 entries per library
    xyz, conn = cube_mesh(n)
    d = host.Db()
    ne = conn.shape[0]
    d.set_mesh(np.arange(1, xyz.shape[0] + 1), xyz, np.arange(1, ne + 1), np.ones(ne), conn + 1, "HEX8_G2")
    d.add_material(1, "Steel", 210000.0, 0.3); d.assign_part(1, 1, "HEX8_G2")
    spc, ld, f = cube_bcs(n)
    d.add_bc(1, "fix", "SPC", spc + 1, np.ones((len(spc), 3)))
    d.add_bc(2, "load", "PointLoad", ld + 1, np.tile(f, (len(ld), 1)))
    d.set_analysis(tol=1e-6)
    rng = np.random.default_rng(5)
    disp, strain, stress = rng.standard_normal((xyz.shape[0], 3)), rng.standard_normal((ne, 8, 6)), rng.standard_normal((ne, 8, 6))
    plain = str(tmp_path / "file.STdb")
    d.write_stdb_with_results(plain, disp, strain, stress)
    fifo = str(tmp_path / "pipe.STdb")
    os.mkfifo(fifo)
    got = []
    t = threading.Thread(target=lambda: got.append(open(fifo, "rb").read()))
    t.start()
    d.write_stdb_with_results(fifo, disp, strain, stress)
    t.join(60)
    assert not t.is_alive() and len(got[0]) > 4_000_000 and got[0] == open(plain, "rb").read()
