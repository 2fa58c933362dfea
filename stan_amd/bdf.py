"""Writes a structured cube as a Nastran short-format .bdf in the layout README.md:35-49 shows
(8-character fixed fields, CHEXA with '+' continuation).  Test/bench input generator; the
reference has none."""
import numpy as np


def _f8(v):
    """a float in at most 8 characters, Nastran short-field style ("7.11-15" for 7.11e-15)"""
    if v == 0:
        return "0."
    a = abs(v)
    if 1e-3 <= a < 1e5:
        s = ("%.6f" % v)[:8]
        s = s.rstrip("0") if "." in s else s + "."
        return s
    m, e = ("%.2e" % v).split("e")
    return (m + "%+d" % int(e))[:8]


def write_bdf(path, xyz, conn, pid=1, first_id=1):
    xyz = np.asarray(xyz)
    conn = np.asarray(conn)
    with open(path, "w") as f:
        f.write("$$  GRID Data\n")
        for i, p in enumerate(xyz):
            f.write("GRID    %8d%8s%8s%8s%8s\n" % (first_id + i, "", _f8(p[0]), _f8(p[1]), _f8(p[2])))
        f.write("$$  CHEXA Elements: First Order\n")
        for e, c in enumerate(conn):
            ids = [first_id + int(x) for x in c]
            f.write("CHEXA   %8d%8d%8d%8d%8d%8d%8d%8d+       \n" % tuple([first_id + e, pid] + ids[:6]))
            f.write("+       %8d%8d\n" % tuple(ids[6:]))
