"""The STdb codec (stan_amd/host/stdb.cpp) pinned against google.protobuf, an independent
implementation of the wire format (VERDICT r01 missing #4; SolverFunctions.cs:48-63; SURVEY.md
App. A).  tests/golden/stdb_golden.npz was produced by tests/golden/make_stdb_golden.py, which
required google.protobuf to decode this codec's bytes to the values that went in and to re-encode
them byte-identically; here the committed bytes are the fixture."""
import importlib.util
import json
import os

import numpy as np
import pytest

from stan_amd import host

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "stdb_golden.npz"))
spec = importlib.util.spec_from_file_location(
    "make_stdb_golden", os.path.join(os.path.dirname(__file__), "golden", "make_stdb_golden.py"))


def _golden_db():
    pytest.importorskip("google.protobuf")   # the generator module imports it at the top
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.golden_db()


def test_codec_reproduces_the_committed_bytes(built_libs):
    d = _golden_db()[0]
    assert d.serialize(packed=False) == GOLD["unpacked"].tobytes()
    assert d.serialize(packed=True) == GOLD["packed"].tobytes()


def test_reader_on_the_committed_bytes(built_libs):
    for key, packed in (("unpacked", False), ("packed", True)):
        b = GOLD[key].tobytes()
        d = host.Db.parse_stdb(b)
        assert d.serialize(packed=packed) == b
        s = d.sizes()
        assert (s["nodes"], s["elements"], s["materials"], s["bcs"], s["nDOF"], s["result_step"]) == (27, 8, 1, 2, 81, 1)
        a = d.analysis()
        assert a["lin_solver"] == "CG" and a["max_iter"] == -3 and a["tol"] == 1e-6
    # both encodings hold the same database
    assert host.Db.parse_stdb(GOLD["packed"].tobytes()).serialize() == GOLD["unpacked"].tobytes()


def test_third_party_protobuf_decodes_and_reencodes_identically(built_libs):
    pytest.importorskip("google.protobuf")
    from google.protobuf import json_format
    from tests import stdb_schema
    want = json.loads(GOLD["decoded_json"].tobytes().decode())
    for key, packed in (("unpacked", False), ("packed", True)):
        b = GOLD[key].tobytes()
        m = stdb_schema.build(packed)()
        m.ParseFromString(b)
        got = json.loads(json.dumps(json_format.MessageToDict(m, preserving_proto_field_name=True), sort_keys=True))
        assert got == want
        assert m.SerializeToString(deterministic=True) == b
    # and a message BUILT by the third party (not a re-encoding of our own bytes) is read back
    DB = stdb_schema.build(False)
    m = DB()
    e = m.NodeLib.add(); e.key = 7; e.value.ID = 7; e.value.Y = -2.5; e.value.DOF.extend([0, 1, 2]); e.value.DispX.append(0.0)
    e = m.NodeLib.add(); e.key = 3; e.value.ID = 3; e.value.X = 1.0; e.value.DOF.extend([3, 4, 5])
    mt = m.MatLib.add(); mt.key = 2; mt.value.ID = 2; mt.value.Type = "Elastic"; mt.value.Name = "Al"; mt.value.E = 70000.0; mt.value.Poisson = 0.33
    m.nDOF = 6
    m.AnalysisLib.Type = "Linear_Statics"; m.AnalysisLib.LinSolver = "CG"; m.AnalysisLib.LinSolverIterMax = -1
    b = m.SerializeToString(deterministic=True)
    d = host.Db.parse_stdb(b)
    assert d.serialize() == b                       # wire order of the dictionary (7 before 3) is kept
    s = d.sizes()
    assert (s["nodes"], s["materials"], s["nDOF"]) == (2, 1, 6) and d.analysis()["max_iter"] == -1
