"""K-nearest template vertices and the SMPL neighbour-blended LBS quantities.

Drop-ins, with the reference's names, argument order and return values, for

    pytorch3d.ops.knn_points(p1, p2, K=K)                    used at /root/reference/hugs/models/hugs_wo_trimlp.py:60,99
    smpl_lbsweight_top_k(lbs_weights, points, template_points, K=6)             hugs_wo_trimlp.py:88-119
    smpl_lbsmap_top_k(lbs_weights, verts_transform, points, template_points, K=6, addition_info=None)    :47-85
    simple_knn._C.distCUDA2(points)                         used at /root/reference/hugs/models/scene.py:20,181  (row f-4)

The search (and, for smpl_lbsweight_top_k -- the one on the every-training-step path, hugs_trimlp.py:318,480 -- the whole
function) runs in hand-written HIP (csrc/knn.hip) behind the C ABI; smpl_lbsmap_top_k keeps the reference's torch
statements after the search because gradients flow through `verts_transform` there.  No CPU fallback.
"""
import ctypes as C
from collections import namedtuple

import torch

from diff_gaussian_rasterization import _load, _raise_last, _require_gpu, _stream_ptr

_KNN = namedtuple("KNN", "dists idx knn")   # pytorch3d's return type (knn is None unless return_nn=True)


def _prep(t, name):
    _require_gpu(t, name)
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def knn_points(p1, p2, lengths1=None, lengths2=None, norm=2, K=1, version=-1, return_nn=False, return_sorted=True):
    """p1 [B,n,3], p2 [B,m,3] -> KNN(dists [B,n,K] squared L2 ascending, idx [B,n,K] int64, knn)."""
    if lengths1 is not None or lengths2 is not None or norm != 2:
        raise NotImplementedError("knn_points (MI355X): only full-length clouds and norm=2 (all the reference uses)")
    if p1.ndim != 3 or p2.ndim != 3 or p1.shape[0] != p2.shape[0] or p1.shape[2] != 3 or p2.shape[2] != 3:
        raise ValueError("knn_points: expected p1 [B,n,3] and p2 [B,m,3]")
    lib = _load()
    lib.hgs_knn_points.restype = C.c_int32
    lib.hgs_knn_points.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    a, b = _prep(p1, "p1"), _prep(p2, "p2")
    B, n, m = a.shape[0], a.shape[1], b.shape[1]
    dists = torch.empty(B, n, K, dtype=torch.float32, device=a.device)
    idx = torch.empty(B, n, K, dtype=torch.int64, device=a.device)
    with torch.cuda.device(a.device):
        for i in range(B):
            rc = lib.hgs_knn_points(n, a[i].data_ptr(), m, b[i].data_ptr(), K, dists[i].data_ptr(), idx[i].data_ptr(),
                                    _stream_ptr(a.device))
            if rc < 0:
                _raise_last(lib, "knn_points")
    nn = None
    if return_nn:
        nn = torch.gather(p2[:, :, None, :].expand(-1, -1, K, -1), 1, idx[..., None].expand(-1, -1, -1, 3))
    return _KNN(dists, idx, nn)


def smpl_lbsweight_top_k(lbs_weights, points, template_points, K=6):
    """-> (xyz_dist [B,n,1], xyz_neighbs_lbs_weight [B,n,J]); one fused kernel per batch element, no autograd (the
    reference's call sites wrap it in torch.no_grad() and its inputs' gradients are cut by the no_grad search)."""
    lib = _load()
    lib.hgs_smpl_lbsweight_top_k.restype = C.c_int32
    lib.hgs_smpl_lbsweight_top_k.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                             C.c_void_p, C.c_void_p, C.c_void_p]
    if points.ndim != 3 or template_points.ndim != 3 or lbs_weights.ndim != 2:
        raise ValueError("smpl_lbsweight_top_k: expected points [B,n,3], template_points [B,m,3], lbs_weights [m,J]")
    p, t, w = _prep(points, "points"), _prep(template_points, "template_points"), _prep(lbs_weights, "lbs_weights")
    B, n, m, J = p.shape[0], p.shape[1], t.shape[1], w.shape[1]
    if w.shape[0] != m:
        raise ValueError("smpl_lbsweight_top_k: lbs_weights must have one row per template point")
    dist = torch.empty(B, n, 1, dtype=torch.float32, device=p.device)
    out = torch.empty(B, n, J, dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        for i in range(B):
            rc = lib.hgs_smpl_lbsweight_top_k(n, p[i].data_ptr(), m, t[i].data_ptr(), w.data_ptr(), J, K, dist[i].data_ptr(),
                                              out[i].data_ptr(), _stream_ptr(p.device))
            if rc < 0:
                _raise_last(lib, "smpl_lbsweight_top_k")
    return dist, out


def batch_index_select(data, inds):
    """hugs_wo_trimlp.py:39-44."""
    bs, nv = data.shape[:2]
    inds = inds + (torch.arange(bs, dtype=torch.int32, device=data.device) * nv)[:, None, None]
    data = data.reshape(bs * nv, *data.shape[2:])
    return data[inds.long()]


def smpl_lbsmap_top_k(lbs_weights, verts_transform, points, template_points, K=6, addition_info=None):
    """-> (xyz_dist, xyz_transform[, xyz_info]); HIP search, then the reference's statements (differentiable in
    verts_transform / addition_info / lbs_weights exactly as upstream)."""
    with torch.no_grad():
        results = knn_points(points, template_points, K=K)
        neighbs_dist, neighbs = results.dists, results.idx
    weight_std2 = 2. * 0.1 ** 2
    nb_w = lbs_weights[neighbs]
    conf = torch.exp(-torch.sum(torch.abs(nb_w - nb_w[..., 0:1, :]), dim=-1) / weight_std2)
    conf = torch.gt(conf, 0.9).float()
    wgt = torch.exp(-neighbs_dist)
    wgt = wgt * conf
    wgt = wgt / wgt.sum(-1, keepdim=True)
    nb_T = batch_index_select(verts_transform, neighbs)
    xyz_transform = torch.sum(wgt.unsqueeze(-1).unsqueeze(-1) * nb_T, dim=2)
    xyz_dist = torch.sum(wgt * neighbs_dist, dim=2, keepdim=True)
    if addition_info is not None:
        nb_info = batch_index_select(addition_info, neighbs)
        return xyz_dist, xyz_transform, torch.sum(wgt.unsqueeze(-1) * nb_info, dim=2)
    return xyz_dist, xyz_transform


def distCUDA2(points):
    """points [n,3] (cuda, fp32) -> [n]: mean squared distance to the three nearest other points (scene.py:181 clamps
    it and takes log(sqrt(.)) as the initial scale).  Exact: a brute-force scan for initialisation-sized clouds, a uniform
    grid search (same bits, O(n)) from 32 768 points on."""
    lib = _load()
    lib.hgs_dist_cuda2_ws.restype = C.c_int32
    lib.hgs_dist_cuda2_ws.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hgs_dist_cuda2_workspace.restype = C.c_size_t
    lib.hgs_dist_cuda2_workspace.argtypes = [C.c_int32]
    if points.ndim != 2 or points.shape[1] != 3:
        raise ValueError("distCUDA2: expected points [n,3]")
    p = _prep(points, "points")
    out = torch.empty(p.shape[0], dtype=torch.float32, device=p.device)
    nbytes = lib.hgs_dist_cuda2_workspace(p.shape[0])
    ws = torch.empty(nbytes, dtype=torch.uint8, device=p.device) if nbytes else None
    with torch.cuda.device(p.device):
        rc = lib.hgs_dist_cuda2_ws(p.shape[0], p.data_ptr(), out.data_ptr(), ws.data_ptr() if ws is not None else None,
                                   _stream_ptr(p.device))
    if rc < 0:
        _raise_last(lib, "distCUDA2")
    return out
