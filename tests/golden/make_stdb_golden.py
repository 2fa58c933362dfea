"""Generates tests/golden/stdb_golden.npz (run in the build container, where google.protobuf
7.35.1 is importable):  python tests/golden/make_stdb_golden.py

The STdb codec of libstan_host.so (stan_amd/host/stdb.cpp; SolverFunctions.cs:48-63 calls
protobuf-net's Serializer on the Database graph) is hand-written.  This script pins it against an
independent protobuf implementation: the bytes stan_host_db_serialize produces for a small database
WITH results (unpacked = what protobuf-net writes without IsPacked, and packed) are decoded by
google.protobuf through the SURVEY.md App. A schema (tests/stdb_schema.py), compared field by
field with what went in, re-encoded by google.protobuf and required to come back BYTE-IDENTICAL.
Committed: both byte strings and the decoded content as JSON (the third-party decoder's view).
What this cannot pin: that protobuf-net 3.0.73 makes the same choices App. A assumes (unpacked
repeated scalars, implicit zero defaults) -- no GUI-written .STdb exists in the reference tree."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from google.protobuf import json_format  # noqa: E402
from stan_amd import host  # noqa: E402
from stan_amd.cube import cube_bcs, cube_mesh  # noqa: E402
from tests import stdb_schema  # noqa: E402


def golden_db():
    """2^3 cube, one node moved off the lattice, material, part, SPC + PointLoad, analysis with a
    NEGATIVE int (10-byte varint), DOFs assigned, synthetic results for increment 1."""
    n = 2
    xyz, conn = cube_mesh(n)
    xyz[13] += [0.125, -0.25, 0.0625]
    d = host.Db()
    ne = conn.shape[0]
    d.set_mesh(np.arange(1, xyz.shape[0] + 1), xyz, np.arange(1, ne + 1), np.ones(ne), conn + 1, "HEX8_G2")
    d.add_material(1, "Steel", 210000.0, 0.3)
    d.assign_part(1, 1, "HEX8_G2")
    spc, ld, f = cube_bcs(n)
    d.add_bc(1, "fix", "SPC", spc + 1, np.ones((len(spc), 3)))
    d.add_bc(2, "load", "PointLoad", ld + 1, np.tile(f, (len(ld), 1)))
    d.set_analysis(tol=1e-6, max_iter=-3)
    d.assign_dof()
    rng = np.random.default_rng(20261002)
    disp = rng.standard_normal((xyz.shape[0], 3))
    disp[spc] = 0.0
    d.set_results(disp, rng.standard_normal((ne, 48)), rng.standard_normal((ne, 48)))
    return d, xyz, conn, disp


def main():
    d, xyz, conn, disp = golden_db()
    out = {}
    for packed in (False, True):
        b = d.serialize(packed=packed)
        DB = stdb_schema.build(packed)
        m = DB()
        m.ParseFromString(b)
        # field-by-field against what went in
        assert [e.key for e in m.NodeLib] == list(range(1, 28)) and m.nDOF == 81
        for i, e in enumerate(m.NodeLib):
            assert (e.value.X, e.value.Y, e.value.Z) == tuple(xyz[i]) and e.value.ID == i + 1
            assert list(e.value.DispX) == [0.0, disp[i, 0]] and len(e.value.DOF) == 3
        for i, e in enumerate(m.ElemLib):
            assert list(e.value.NList) == list(conn[i] + 1) and e.value.Type == "HEX8_G2" and e.value.MatID == 1
            assert len(e.value.Strain) == 2 and (e.value.Strain[1].Rows, e.value.Strain[1].Cols) == (8, 6)
        assert m.AnalysisLib.LinSolverIterMax == -3 and m.AnalysisLib.Result_StepNo == 1
        assert m.MatLib[0].value.E == 210000.0 and m.BCLib[1].value.Type == "PointLoad"
        b2 = m.SerializeToString(deterministic=True)
        assert b2 == b, "google.protobuf re-encodes the database differently"
        assert host.Db.parse_stdb(b2).serialize(packed=packed) == b
        out["packed" if packed else "unpacked"] = np.frombuffer(b, dtype=np.uint8)
        if not packed:
            out["decoded_json"] = np.frombuffer(
                json.dumps(json_format.MessageToDict(m, preserving_proto_field_name=True), sort_keys=True).encode(),
                dtype=np.uint8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "stdb_golden.npz"), **out)
    print("wrote stdb_golden.npz: %d B unpacked, %d B packed" % (out["unpacked"].size, out["packed"].size))


if __name__ == "__main__":
    main()
