"""Independent numpy restatement used only to cross-check the oracle (tests only)."""
import numpy as np

SX, SY, SZ = 0x66, 0xCC, 0xF0


def _sg(m, i):
    return 1.0 if (m >> i) & 1 else -1.0


def dnl(i, p):
    sx, sy, sz = _sg(SX, i), _sg(SY, i), _sg(SZ, i)
    fx, fy, fz = 1 + sx * p[0], 1 + sy * p[1], 1 + sz * p[2]
    return np.array([0.125 * sx * fy * fz, 0.125 * sy * fx * fz, 0.125 * sz * fx * fy])


def D_matrix(E, nu):
    lam = E * nu / ((1 - 2 * nu) * (1 + nu))
    G = 0.5 * E / (1 + nu)
    D = np.zeros((6, 6))
    D[:3, :3] = lam
    D[0, 0] = D[1, 1] = D[2, 2] = lam + 2 * G
    D[3, 3] = D[4, 4] = D[5, 5] = G
    return D


def ke_numpy(x, E, nu, etype):
    """K_e = sum_g B^T D B detJ w with np.linalg (strain order xx,yy,zz,xy,yz,xz)."""
    D = D_matrix(E, nu)
    gl = np.sqrt(1.0 / 3.0) if etype == 2 else 0.0
    K = np.zeros((24, 24))
    for g in range(8 if etype == 2 else 1):
        w = 1.0 if etype == 2 else 8.0
        p = np.array([_sg(SX, g), _sg(SY, g), _sg(SZ, g)]) * gl
        dN = np.stack([dnl(i, p) for i in range(8)], 1)
        J = dN @ x
        gr = np.linalg.solve(J, dN)
        B = np.zeros((6, 24))
        for i in range(8):
            B[0, 3 * i] = gr[0, i]
            B[1, 3 * i + 1] = gr[1, i]
            B[2, 3 * i + 2] = gr[2, i]
            B[3, 3 * i], B[3, 3 * i + 1] = gr[1, i], gr[0, i]
            B[4, 3 * i + 1], B[4, 3 * i + 2] = gr[2, i], gr[1, i]
            B[5, 3 * i], B[5, 3 * i + 2] = gr[2, i], gr[0, i]
        K += B.T @ D @ B * np.linalg.det(J) * w
    return K


UNIT = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0],
                 [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], float)


def random_hexes(n, seed=7, jitter=0.2):
    rng = np.random.default_rng(seed)
    scale = rng.uniform(0.5, 2.5, (n, 1, 3))
    return UNIT[None] * scale + rng.uniform(-jitter, jitter, (n, 8, 3)) * scale.min(axis=2, keepdims=True)
