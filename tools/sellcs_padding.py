"""Padding of BSELL-64 off the cube, as a function of the sorting window sigma (SELL-C-sigma):
a box of hexes with a fraction of its elements knocked out (the fuzz generator's mesh classes,
tests/fuzz.py), rows in reference (AssignDOF) order.  CPU only: python tools/sellcs_padding.py [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
from stan_amd import host
from stan_amd.cube import cube_mesh


from stan_amd.cube import perforated_mesh as perforated


def rowlens(n_nodes, conn, idx):
    c = idx[conn]                                  # block rows of the element's nodes
    a = np.repeat(c, 8, axis=1).ravel()
    b = np.tile(c, (1, 8)).ravel()
    key = np.unique(a.astype(np.int64) * n_nodes + b)
    return np.bincount((key // n_nodes).astype(np.int64), minlength=n_nodes)


def padding(rl, sigma):
    n = rl.shape[0]
    npad = (n + 63) // 64 * 64
    r = np.zeros(npad, dtype=np.int64); r[:n] = rl
    w = 64 * sigma
    slots = 0
    for s in range(0, npad, w):
        win = np.sort(r[s:s + w])[::-1] if sigma > 1 else r[s:s + w]
        m = win.shape[0] // 64
        slots += win[:m * 64].reshape(m, 64).max(axis=1).sum() * 64
    return slots / rl.sum() - 1.0


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    for frac in (0.0, 0.15, 0.4):
        xyz, conn = perforated(n, frac)
        idx, dof = host.assign_dof(xyz.shape[0], conn)
        rl = rowlens(xyz.shape[0], conn, idx)
        print("%d^3 box, %2.0f %% of the elements knocked out: %d nodes, %d blocks; padding  " % (n, 100 * frac, xyz.shape[0], rl.sum()) +
              "  ".join("sigma=%d: %.1f %%" % (s, 100 * padding(rl, s)) for s in (1, 2, 4, 8, 16, 32, 64)), flush=True)
