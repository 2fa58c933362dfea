"""CPU simulation of the packed column stream (k_pack_cols, cg_setup_kernels.inc) on the n^3 cube in AssignDOF order: which
fraction of the ELL slots fits 16-bit offsets with ONE base per slot (round 2; padding entries = the row's own column
included: 'one'; padding excluded: 'one_nopad'), with TWO bases per slot -- the slice's full-width rows / its shorter rows
(round 4, mode 2: 'two') -- and with one base per row length ('multi').  No GPU needed.
usage: python tools/packed_columns_sim.py n      (148: 0.984 -> 0.9991; 200: 0.6385 -> 0.9983; profiles/r04/packed_columns_simulation.txt)"""
import sys, time
sys.path.insert(0,'/root/repo')
import numpy as np
from stan_amd import host
from stan_amd.cube import cube_mesh
n = int(sys.argv[1])
m = n+1
xyz, conn = cube_mesh(n)
node_index, node_dof = host.assign_dof(m**3, conn)
del xyz, conn
row_of_node = (node_dof.reshape(-1,3)[:,0]//3).astype(np.int64)
node_of_row = np.empty(m**3, np.int64); node_of_row[row_of_node] = np.arange(m**3)
nrows = m**3
nsl = (nrows+63)//64
BIG = np.int64(1)<<40
tot = dict(slots=0, one=0, one_nopad=0, two=0, multi=0)
offs = [(a,b,c) for c in (-1,0,1) for b in (-1,0,1) for a in (-1,0,1)]
CH = 4096   # slices per chunk
for s0 in range(0, nsl, CH):
    s1 = min(nsl, s0+CH)
    r0, r1 = s0*64, min(nrows, s1*64)
    nodes = node_of_row[r0:r1]
    i = nodes % m; j = (nodes//m) % m; k = nodes//(m*m)
    cols = np.full((s1*64 - r0, 27), BIG, np.int64)
    for q,(a,b,c) in enumerate(offs):
        ii, jj, kk = i+a, j+b, k+c
        ok = (ii>=0)&(ii<m)&(jj>=0)&(jj<m)&(kk>=0)&(kk<m)
        nb = ii + m*(jj + m*kk)
        cols[:r1-r0, q] = np.where(ok, row_of_node[np.where(ok, nb, 0)], BIG)
    cols.sort(axis=1)
    ln = (cols < BIG).sum(axis=1)           # row lengths (0 for padding rows beyond nrows)
    cols = cols.reshape(-1, 64, 27); ln = ln.reshape(-1, 64)
    w = ln.max(axis=1)                      # slice width
    own = (np.arange(r0, s1*64)).reshape(-1,64)
    live = np.arange(27)[None,None,:] < ln[:,:,None]
    inw = np.arange(27)[None,:] < w[:,None]              # slot exists in slice
    # scheme 0: current (padding = own column)
    c0 = np.where(live, cols, own[:,:,None])
    spread0 = c0.max(axis=1) - c0.min(axis=1)
    ok0 = ((spread0 < 65536) | ~inw).all(axis=1)
    # scheme 1: padding excluded
    mx = np.where(live, cols, -1).max(axis=1); mn = np.where(live, cols, BIG).min(axis=1)
    ok1 = (((mx-mn) < 65536) | ~inw).all(axis=1)
    # scheme 2: two classes (A: len == w, B: others), padding excluded
    A = (ln == w[:,None])[:,:,None]
    def spread(mask):
        mx = np.where(mask, cols, -1).max(axis=1); mn = np.where(mask, cols, BIG).min(axis=1)
        return np.where(mx>=0, mx-mn, 0)
    ok2 = (((spread(live&A) < 65536) & (spread(live&~A) < 65536)) | ~inw).all(axis=1)
    # scheme 3: one class per distinct length
    ok3 = np.ones(len(w), bool)
    for L in (27,18,12,8):
        ok3 &= ((spread(live & (ln==L)[:,:,None]) < 65536) | ~inw).all(axis=1)
    tot['slots'] += int(w.sum()); tot['one'] += int(w[ok0].sum()); tot['one_nopad'] += int(w[ok1].sum())
    tot['two'] += int(w[ok2].sum()); tot['multi'] += int(w[ok3].sum())
print(n, {k: (v, round(v/tot['slots'],4)) for k,v in tot.items()})
