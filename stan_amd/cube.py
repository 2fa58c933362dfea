"""Synthetic structured HEX8 cube (SURVEY.md Appendix E conventions).

The reference has no mesh generator; this is the workload generator used by
bench.py and the tests.  Node IDs 1..(n+1)^3 x-fastest, element IDs 1..n^3
x-fastest, CHEXA node order = the natural-coordinate sign table of
FE_Library.cs:225-235, so det J = h^3/8 > 0.
"""
import numpy as np


def cube_mesh(n, h=1.0, jitter=0.0, seed=12345):
    """Returns xyz [(n+1)^3, 3] float64 and conn [n^3, 8] int32 of node *indices*
    (NodeLib order == ID order, index = ID - 1)."""
    m = n + 1
    # (broadcasting, not meshgrid: the 400^3 mesh of BASELINE config 5 is 64 M elements -- a minute less of host time)
    ax = np.arange(m, dtype=np.float64) * h
    xyz = np.empty((m, m, m, 3), dtype=np.float64)        # [k][j][i][xyz], x fastest
    xyz[..., 0] = ax[None, None, :]
    xyz[..., 1] = ax[None, :, None]
    xyz[..., 2] = ax[:, None, None]
    xyz = xyz.reshape(-1, 3)
    if jitter:
        rng = np.random.default_rng(seed)
        xyz = xyz + rng.uniform(-jitter * h, jitter * h, size=xyz.shape)
    e = np.arange(n, dtype=np.int32)
    base = (e[None, None, :] + np.int32(m) * (e[None, :, None] + np.int32(m) * e[:, None, None])).reshape(-1)   # node (i, j, k)
    off = np.array([0, 1, 1 + m, m, m * m, 1 + m * m, 1 + m + m * m, m + m * m], dtype=np.int32)
    conn = base[:, None] + off[None, :]
    return xyz, conn


def cube_bcs(n, h=1.0, clamp_faces="x", load=(0.0, 0.0, 50.0)):
    """SPC (1,1,1) on the nodes of the clamped faces (x=0; 'xyz' = x=0 u y=0 u z=0,
    needed for HEX8_G1) and PointLoad `load` on every node with x = n*h
    (README.md:58-69's example values).  Returns
    (spc_nodes int32[], load_nodes int32[], load float64[3]) as node indices."""
    m = n + 1
    idx = np.arange(m ** 3)
    i = idx % m
    j = (idx // m) % m
    k = idx // (m * m)
    fixed = np.zeros(m ** 3, dtype=bool)
    if "x" in clamp_faces:
        fixed |= i == 0
    if "y" in clamp_faces:
        fixed |= j == 0
    if "z" in clamp_faces:
        fixed |= k == 0
    spc = idx[fixed].astype(np.int32)
    ld = idx[i == n].astype(np.int32)
    return spc, ld, np.asarray(load, dtype=np.float64)


def star_mesh(k=5, layers=3, rings=2, h=1.0):
    """An UNSTRUCTURED hex mesh: a regular k-gon cut into k quadrilateral sectors around its
    centre (each sector refined `rings` x `rings`), extruded `layers` times in z.  The centre
    line has valence k in the plane (2k incident hexes inside the stack: more than the 8 of a
    structured mesh for k > 4, fewer for k = 3) and the sector seams are irregular too.
    Returns xyz [n,3] float64 and conn [e,8] int32 (CHEXA order, det J > 0)."""
    m = rings
    pts = {}
    coords = []

    def pid(p):
        key = (round(p[0], 9), round(p[1], 9))
        if key not in pts:
            pts[key] = len(coords)
            coords.append((p[0], p[1]))
        return pts[key]

    quads = []
    ang = 2 * np.pi / k
    c = np.zeros(2)
    for s in range(k):
        v = np.array([np.cos(s * ang), np.sin(s * ang)])                 # polygon vertex
        e0 = 0.5 * (v + np.array([np.cos((s - 1) * ang), np.sin((s - 1) * ang)]))  # mid of previous edge
        e1 = 0.5 * (v + np.array([np.cos((s + 1) * ang), np.sin((s + 1) * ang)]))  # mid of next edge
        # sector quad (counter-clockwise): centre -> e0 -> v -> e1, bilinear m x m refinement
        def P(a, b):
            u, w = a / m, b / m
            return (1 - u) * (1 - w) * c + u * (1 - w) * e0 + u * w * v + (1 - u) * w * e1
        for a in range(m):
            for b in range(m):
                quads.append([pid(P(a, b)), pid(P(a + 1, b)), pid(P(a + 1, b + 1)), pid(P(a, b + 1))])
    n2 = len(coords)
    xy = np.array(coords)
    xyz = np.concatenate([np.column_stack([xy, np.full(n2, z * h)]) for z in range(layers + 1)])
    conn = []
    for z in range(layers):
        for q in quads:
            conn.append([q[0] + z * n2, q[1] + z * n2, q[2] + z * n2, q[3] + z * n2,
                         q[0] + (z + 1) * n2, q[1] + (z + 1) * n2, q[2] + (z + 1) * n2,
                         q[3] + (z + 1) * n2])
    return xyz.astype(np.float64), np.array(conn, dtype=np.int32)


def perforated_mesh(n, frac, seed=7):
    """An IRREGULAR test mesh at a chosen size: the n^3 cube with a fraction `frac` of its elements knocked
    out at random (the mesh class of tests/fuzz.py), largest connected component kept, unused nodes dropped.
    Row lengths of K then vary from node to node (Database.ReadNastranMesh admits any CHEXA mesh,
    Database.cs:39-111).  Returns xyz [n_nodes,3] float64 and conn [n_elem,8] int32."""
    import scipy.sparse as sp
    import scipy.sparse.csgraph
    xyz, conn = cube_mesh(n)
    rng = np.random.default_rng(seed)
    conn = conn[rng.random(conn.shape[0]) >= frac]
    ne = conn.shape[0]
    inc = sp.csr_matrix((np.ones(ne * 8, dtype=np.int8), (np.repeat(np.arange(ne), 8), conn.ravel())),
                        shape=(ne, xyz.shape[0]))
    _, lab = sp.csgraph.connected_components((inc @ inc.T).tocsr(), directed=False)
    conn = conn[lab == np.argmax(np.bincount(lab))]
    used = np.unique(conn)
    new = np.full(xyz.shape[0], -1, dtype=np.int64)
    new[used] = np.arange(used.shape[0])
    return xyz[used], new[conn].astype(np.int32)


def revolved_mesh(sectors=72, rings=2, layers=3, r0=1.0, h=1.0):
    """A solid of revolution meshed the way pre-processors do it: `sectors` wedge-shaped COLLAPSED hexes around
    the axis (CHEXA with node 4 = node 1 and node 8 = node 5: the axis node is listed twice), `rings` - 1 rings of
    ordinary hexes outside them, `layers` layers in z.  An interior axis node belongs to 2 * sectors elements with
    2 incidences each (72 sectors: 288 (element, local node) pairs) and couples to 3 * (sectors + 1) nodes -- the
    reference accepts any of it (Node.RemoveElemDuplicates, Node.cs:202-205; Database.cs:149-176 puts no bound on
    the elements at a node).  Returns xyz [n,3] float64 and conn [e,8] int32 (det J > 0 at every Gauss point)."""
    ang = 2.0 * np.pi * np.arange(sectors) / sectors
    per_layer = 1 + sectors * rings      # axis node, then ring 1 .. rings (sector fastest)
    xyz = []
    for z in range(layers + 1):
        xyz.append((0.0, 0.0, z * h))
        for r in range(1, rings + 1):
            for s in range(sectors):
                xyz.append((r0 * r * np.cos(ang[s]), r0 * r * np.sin(ang[s]), z * h))

    def nid(z, r, s):   # r = 0: the axis
        return z * per_layer + (0 if r == 0 else 1 + (r - 1) * sectors + s % sectors)

    conn = []
    for z in range(layers):
        for s in range(sectors):
            a0, a1 = nid(z, 0, 0), nid(z + 1, 0, 0)
            conn.append([a0, nid(z, 1, s), nid(z, 1, s + 1), a0, a1, nid(z + 1, 1, s), nid(z + 1, 1, s + 1), a1])
            for r in range(1, rings):
                conn.append([nid(z, r, s), nid(z, r + 1, s), nid(z, r + 1, s + 1), nid(z, r, s + 1),
                             nid(z + 1, r, s), nid(z + 1, r + 1, s), nid(z + 1, r + 1, s + 1), nid(z + 1, r, s + 1)])
    return np.array(xyz, dtype=np.float64), np.array(conn, dtype=np.int32)
