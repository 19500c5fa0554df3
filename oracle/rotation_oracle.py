"""CPU oracle for row f-7 (test infrastructure only -- never imported by the product path): rotation_6d_to_matrix and
matrix_to_quaternion of /root/reference/hugs/utils/rotations.py:552-573,94-156 restated in numpy float64, with the backward
autograd derives from those statements.  Pinned by tests/golden/reference_rotations.npz (the reference's own functions compiled
from its source and run on CPU with autograd, tests/golden/make_golden_rotations.py)."""
import numpy as np

_ROWS = {  # candidate row b, column j != b: (index of the + term, index of the second term, its sign) in the flattened 3x3 (:136-144)
    (0, 1): (7, 5, -1), (0, 2): (2, 6, -1), (0, 3): (3, 1, -1), (1, 2): (3, 1, 1), (1, 3): (2, 6, 1), (2, 3): (5, 7, 1)}
_SIGNS = np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], np.float64)   # t_b = 1 + s . (m00, m11, m22)  (:123-133)


def _select(m):
    t = 1.0 + m[:, [0, 4, 8]] @ _SIGNS.T
    q_abs = np.where(t > 0, np.sqrt(np.maximum(t, 0)), 0.0)                        # _sqrt_positive_part (:94-102)
    return t, q_abs, q_abs.argmax(1)                                               # first maximum (:154-156)


def matrix_to_quaternion(matrix):
    m = np.asarray(matrix, np.float64).reshape(-1, 9)
    t, q_abs, b = _select(m)
    out = np.empty((m.shape[0], 4))
    for i in range(m.shape[0]):
        bi = b[i]
        for j in range(4):
            if j == bi:
                out[i, j] = q_abs[i, bi] ** 2
            else:
                p, q, s = _ROWS[(min(bi, j), max(bi, j))]
                out[i, j] = m[i, p] + s * m[i, q]
        out[i] /= 2.0 * max(q_abs[i, bi], 0.1)                                     # :148-149
    return out.reshape(np.shape(matrix)[:-2] + (4,))


def matrix_to_quaternion_backward(matrix, g):
    m = np.asarray(matrix, np.float64).reshape(-1, 9)
    g = np.asarray(g, np.float64).reshape(-1, 4)
    t, q_abs, b = _select(m)
    d = np.zeros_like(m)
    for i in range(m.shape[0]):
        bi = b[i]
        D = max(q_abs[i, bi], 0.1)
        gD, gt = 0.0, 0.0
        for j in range(4):
            gN = g[i, j] / (2 * D)
            if j == bi:
                N = q_abs[i, bi] ** 2
                gt += gN if t[i, bi] > 0 else 0.0
            else:
                p, q, s = _ROWS[(min(bi, j), max(bi, j))]
                N = m[i, p] + s * m[i, q]
                d[i, p] += gN
                d[i, q] += s * gN
            gD -= g[i, j] * N / (2 * D * D)
        if q_abs[i, bi] > 0.1:
            gt += gD / (2 * q_abs[i, bi])
        d[i, [0, 4, 8]] += gt * _SIGNS[bi]
    return d.reshape(np.shape(matrix))


def _normalize(x):
    n = np.linalg.norm(x, axis=-1, keepdims=True)
    return x / np.maximum(n, 1e-12), n


def rotation_6d_to_matrix(d6):
    d6 = np.asarray(d6, np.float64)
    b1, _ = _normalize(d6[..., :3])
    u = d6[..., 3:] - (b1 * d6[..., 3:]).sum(-1, keepdims=True) * b1
    b2, _ = _normalize(u)
    return np.stack([b1, b2, np.cross(b1, b2)], axis=-2)


def _normalize_backward(y, n, g):
    return np.where(n > 1e-12, (g - y * (y * g).sum(-1, keepdims=True)) / np.maximum(n, 1e-12), g / 1e-12)


def rotation_6d_to_matrix_backward(d6, G):
    d6, G = np.asarray(d6, np.float64), np.asarray(G, np.float64)
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1, n1 = _normalize(a1)
    s = (b1 * a2).sum(-1, keepdims=True)
    u = a2 - s * b1
    b2, nu = _normalize(u)
    g1, g2, g3 = G[..., 0, :], G[..., 1, :], G[..., 2, :]
    gb1 = g1 + np.cross(b2, g3)
    gb2 = g2 + np.cross(g3, b1)
    gu = _normalize_backward(b2, nu, gb2)
    gub1 = (gu * b1).sum(-1, keepdims=True)
    ga2 = gu - gub1 * b1
    gb1 = gb1 - s * gu - gub1 * a2
    return np.concatenate([_normalize_backward(b1, n1, gb1), ga2], axis=-1)
