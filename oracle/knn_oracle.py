"""CPU oracle for SURVEY.md 8f row f-2 (test infrastructure only -- never imported by the product path).

knn_points: the reference calls pytorch3d.ops.knn_points (/root/reference/hugs/models/hugs_wo_trimlp.py:12,60,99), a pip
dependency that is NOT in /root/reference and not installed here: PARITY UNPINNED for the search itself.  Restated from
its published contract (pytorch3d/ops/knn.py docstring): for every query the K template points with the smallest
squared Euclidean distance, ascending, int64 indices.  Chosen tie rule: equal distances keep the lower index first.
Distances are accumulated per dimension in fp32, ((dx^2 + dy^2) + dz^2), which numpy float32 reproduces bit for bit.

smpl_lbsweight_top_k / smpl_lbsmap_top_k: restatements of hugs_wo_trimlp.py:88-119 and :47-85 (every statement after the
search), pinned by golden vectors produced by the reference's own statements (tests/golden/make_golden.py compiles
the two functions from the reference's source file and feeds them this module's knn_points).
"""
import numpy as np


def knn_points(points, template_points, K):
    """points [n,3], template_points [m,3] float32 -> (dists [n,K] float32 ascending, idx [n,K] int64)."""
    p = np.ascontiguousarray(points, dtype=np.float32)
    t = np.ascontiguousarray(template_points, dtype=np.float32)
    n, m = p.shape[0], t.shape[0]
    assert 1 <= K <= m
    dists = np.empty((n, K), np.float32)
    idx = np.empty((n, K), np.int64)
    step = max(1, (1 << 24) // max(m, 1))
    for s in range(0, n, step):
        q = p[s:s + step]
        dx = q[:, None, 0] - t[None, :, 0]
        dy = q[:, None, 1] - t[None, :, 1]
        dz = q[:, None, 2] - t[None, :, 2]
        d = (dx * dx + dy * dy) + dz * dz                      # fp32 throughout, this association
        order = np.argsort(d, axis=1, kind="stable")[:, :K]    # stable: ties -> lower index first
        idx[s:s + step] = order
        dists[s:s + step] = np.take_along_axis(d, order, axis=1)
    return dists, idx


def _blend_weights(lbs_weights, dists, idx):
    """hugs_wo_trimlp.py:101-113 -- the normalised, confidence-gated neighbour weights [n,K] (fp32)."""
    w = np.ascontiguousarray(lbs_weights, dtype=np.float32)
    weight_std2 = 2.0 * 0.1 ** 2                                            # :103-104 (python double)
    nb = w[idx]                                                             # :105  [n,K,J]
    l1 = np.abs(nb - nb[:, 0:1, :]).sum(-1, dtype=np.float32)               # :106-109
    conf = np.exp((-l1 / np.float32(weight_std2)).astype(np.float32)).astype(np.float32)
    conf = (conf > np.float32(0.9)).astype(np.float32)                      # :110
    wgt = np.exp(-dists).astype(np.float32) * conf                          # :111-112
    wgt = wgt / wgt.sum(-1, keepdims=True, dtype=np.float32)                # :113
    return nb, wgt.astype(np.float32)


def smpl_lbsweight_top_k(lbs_weights, points, template_points, K=6):
    """-> (xyz_dist [n,1], weights [n,J]), hugs_wo_trimlp.py:88-119 for batch size 1."""
    dists, idx = knn_points(points, template_points, K)
    nb, wgt = _blend_weights(lbs_weights, dists, idx)
    out_w = (wgt[:, :, None] * nb).sum(1, dtype=np.float32)                 # :116
    xyz_dist = (wgt * dists).sum(1, keepdims=True, dtype=np.float32)        # :117
    return xyz_dist, out_w


def smpl_lbsmap_top_k(lbs_weights, verts_transform, points, template_points, K=6, addition_info=None):
    """-> (xyz_dist [n,1], xyz_transform [n,4,4][, xyz_info]), hugs_wo_trimlp.py:47-85 for batch size 1."""
    dists, idx = knn_points(points, template_points, K)
    _, wgt = _blend_weights(lbs_weights, dists, idx)
    T = np.ascontiguousarray(verts_transform, dtype=np.float32)[idx]        # :76  [n,K,4,4]
    xyz_transform = (wgt[:, :, None, None] * T).sum(1, dtype=np.float32)    # :77
    xyz_dist = (wgt * dists).sum(1, keepdims=True, dtype=np.float32)        # :78
    if addition_info is not None:
        info = np.ascontiguousarray(addition_info, dtype=np.float32)[idx]   # :81
        return xyz_dist, xyz_transform, (wgt[:, :, None] * info).sum(1, dtype=np.float32)
    return xyz_dist, xyz_transform


def dist_cuda2(points):
    """simple_knn._C.distCUDA2 (called at /root/reference/hugs/models/scene.py:181; the simple-knn submodule is NOT in
    /root/reference: PARITY UNPINNED, restated from its published behaviour): for every point the mean of the squared
    distances to its three nearest OTHER points of the same cloud.  points [n,3] float32 -> [n] float32."""
    p = np.ascontiguousarray(points, dtype=np.float32)
    n = p.shape[0]
    assert n >= 4
    out = np.empty(n, np.float32)
    step = max(1, (1 << 24) // n)
    for s in range(0, n, step):
        q = p[s:s + step]
        dx = q[:, None, 0] - p[None, :, 0]
        dy = q[:, None, 1] - p[None, :, 1]
        dz = q[:, None, 2] - p[None, :, 2]
        d = (dx * dx + dy * dy) + dz * dz
        d[np.arange(q.shape[0]), np.arange(s, s + q.shape[0])] = np.inf      # a point is not its own neighbour
        b = np.sort(d, axis=1)[:, :3]
        out[s:s + step] = ((b[:, 0] + b[:, 1]) + b[:, 2]) / np.float32(3.0)
    return out
