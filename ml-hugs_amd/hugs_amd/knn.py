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
import os
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


def _template_workspace(lib, n, m, device):
    """Scratch for the grid over the template (hgs_knn_workspace): with it the searches walk a few cells per point instead
    of scanning the whole template -- same neighbours, same order."""
    lib.hgs_knn_workspace.restype = C.c_size_t
    lib.hgs_knn_workspace.argtypes = [C.c_int32, C.c_int32]
    if os.environ.get("HGS_KNN_GRID", "1") == "0":            # A/B switch: the scan of the whole template
        return None
    nbytes = lib.hgs_knn_workspace(n, m)
    return torch.empty(nbytes, dtype=torch.uint8, device=device) if nbytes else None


def knn_points(p1, p2, lengths1=None, lengths2=None, norm=2, K=1, version=-1, return_nn=False, return_sorted=True):
    """p1 [B,n,3], p2 [B,m,3] -> KNN(dists [B,n,K] squared L2 ascending, idx [B,n,K] int64, knn)."""
    if lengths1 is not None or lengths2 is not None or norm != 2:
        raise NotImplementedError("knn_points (MI355X): only full-length clouds and norm=2 (all the reference uses)")
    if p1.ndim != 3 or p2.ndim != 3 or p1.shape[0] != p2.shape[0] or p1.shape[2] != 3 or p2.shape[2] != 3:
        raise ValueError("knn_points: expected p1 [B,n,3] and p2 [B,m,3]")
    lib = _load()
    lib.hgs_knn_points_ws.restype = C.c_int32
    lib.hgs_knn_points_ws.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    a, b = _prep(p1, "p1"), _prep(p2, "p2")
    B, n, m = a.shape[0], a.shape[1], b.shape[1]
    dists = torch.empty(B, n, K, dtype=torch.float32, device=a.device)
    idx = torch.empty(B, n, K, dtype=torch.int64, device=a.device)
    ws = _template_workspace(lib, n, m, a.device)
    with torch.cuda.device(a.device):
        for i in range(B):
            rc = lib.hgs_knn_points_ws(n, a[i].data_ptr(), m, b[i].data_ptr(), K, dists[i].data_ptr(), idx[i].data_ptr(),
                                       ws.data_ptr() if ws is not None else None, _stream_ptr(a.device))
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
    lib.hgs_smpl_lbsweight_top_k_ws.restype = C.c_int32
    lib.hgs_smpl_lbsweight_top_k_ws.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    if points.ndim != 3 or template_points.ndim != 3 or lbs_weights.ndim != 2:
        raise ValueError("smpl_lbsweight_top_k: expected points [B,n,3], template_points [B,m,3], lbs_weights [m,J]")
    p, t, w = _prep(points, "points"), _prep(template_points, "template_points"), _prep(lbs_weights, "lbs_weights")
    B, n, m, J = p.shape[0], p.shape[1], t.shape[1], w.shape[1]
    if w.shape[0] != m:
        raise ValueError("smpl_lbsweight_top_k: lbs_weights must have one row per template point")
    dist = torch.empty(B, n, 1, dtype=torch.float32, device=p.device)
    out = torch.empty(B, n, J, dtype=torch.float32, device=p.device)
    ws = _template_workspace(lib, n, m, p.device)
    with torch.cuda.device(p.device):
        for i in range(B):
            rc = lib.hgs_smpl_lbsweight_top_k_ws(n, p[i].data_ptr(), m, t[i].data_ptr(), w.data_ptr(), J, K, dist[i].data_ptr(),
                                                 out[i].data_ptr(), ws.data_ptr() if ws is not None else None, _stream_ptr(p.device))
            if rc < 0:
                _raise_last(lib, "smpl_lbsweight_top_k")
    return dist, out


def batch_index_select(data, inds):
    """hugs_wo_trimlp.py:39-44."""
    bs, nv = data.shape[:2]
    inds = inds + (torch.arange(bs, dtype=torch.int32, device=data.device) * nv)[:, None, None]
    data = data.reshape(bs * nv, *data.shape[2:])
    return data[inds.long()]


class _LbsMapTopK(torch.autograd.Function):
    """One batch element of smpl_lbsmap_top_k: hgs_smpl_lbsmap_top_k forward (search + confidence-gated weights + blend of the
    neighbours' 4x4 transforms and optional per-vertex info in one kernel), hgs_smpl_lbsmap_top_k_backward (scatter-add of the
    weighted gradients into verts_transform / addition_info)."""

    @staticmethod
    def forward(ctx, lbs_weights, verts_transform, points, template_points, K, addition_info):
        lib = _load()
        lib.hgs_smpl_lbsmap_top_k.restype = C.c_int32
        lib.hgs_smpl_lbsmap_top_k.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                              C.c_void_p, C.c_int32] + [C.c_void_p] * 7
        p, t, w = _prep(points, "points"), _prep(template_points, "template_points"), _prep(lbs_weights, "lbs_weights")
        vt = _prep(verts_transform.reshape(verts_transform.shape[0], 16), "verts_transform")
        info = _prep(addition_info, "addition_info") if addition_info is not None else None
        n, m, J = p.shape[0], t.shape[0], w.shape[1]
        Cc = info.shape[1] if info is not None else 0
        dev = p.device
        dist = torch.empty(n, 1, dtype=torch.float32, device=dev)
        out_T = torch.empty(n, 4, 4, dtype=torch.float32, device=dev)
        out_info = torch.empty(n, Cc, dtype=torch.float32, device=dev) if info is not None else None
        idx = torch.empty(n, K, dtype=torch.int32, device=dev)
        wgt = torch.empty(n, K, dtype=torch.float32, device=dev)
        ws = _template_workspace(lib, n, m, dev)
        with torch.cuda.device(dev):
            rc = lib.hgs_smpl_lbsmap_top_k(n, p.data_ptr(), m, t.data_ptr(), w.data_ptr(), J, K, vt.data_ptr(),
                                           info.data_ptr() if info is not None else None, Cc, dist.data_ptr(), out_T.data_ptr(),
                                           out_info.data_ptr() if info is not None else None, idx.data_ptr(), wgt.data_ptr(),
                                           ws.data_ptr() if ws is not None else None, _stream_ptr(dev))
        if rc < 0:
            _raise_last(lib, "smpl_lbsmap_top_k")
        ctx.save_for_backward(idx, wgt)
        ctx.dims = (n, K, m, Cc, tuple(verts_transform.shape), tuple(addition_info.shape) if addition_info is not None else None)
        ctx.mark_non_differentiable(dist)
        return (dist, out_T, out_info) if info is not None else (dist, out_T)

    @staticmethod
    def backward(ctx, _g_dist, g_T, g_info=None):
        lib = _load()
        lib.hgs_smpl_lbsmap_top_k_backward.restype = C.c_int32
        lib.hgs_smpl_lbsmap_top_k_backward.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                       C.c_void_p, C.c_void_p, C.c_void_p]
        idx, wgt = ctx.saved_tensors
        n, K, m, Cc, vt_shape, info_shape = ctx.dims
        dev = idx.device
        need_T = g_T is not None and ctx.needs_input_grad[1]
        need_info = g_info is not None and info_shape is not None and ctx.needs_input_grad[5]
        d_vt = torch.zeros(m, 16, dtype=torch.float32, device=dev) if need_T else None
        d_info = torch.zeros(m, Cc, dtype=torch.float32, device=dev) if need_info else None
        if need_T or need_info:
            gT = g_T.contiguous().float() if need_T else None
            gI = g_info.contiguous().float() if need_info else None
            with torch.cuda.device(dev):
                rc = lib.hgs_smpl_lbsmap_top_k_backward(n, K, idx.data_ptr(), wgt.data_ptr(), gT.data_ptr() if need_T else None,
                                                        gI.data_ptr() if need_info else None, Cc,
                                                        d_vt.data_ptr() if need_T else None, d_info.data_ptr() if need_info else None,
                                                        _stream_ptr(dev))
            if rc < 0:
                _raise_last(lib, "smpl_lbsmap_top_k_backward")
        return (None, d_vt.view(vt_shape) if need_T else None, None, None, None, d_info.view(info_shape) if need_info else None)


def smpl_lbsmap_top_k(lbs_weights, verts_transform, points, template_points, K=6, addition_info=None):
    """-> (xyz_dist [B,n,1], xyz_transform [B,n,4,4][, xyz_info [B,n,C]]) of hugs_wo_trimlp.py:47-85: the search, the
    confidence-gated neighbour weights and the blends in ONE kernel per batch element, differentiable in verts_transform and
    addition_info as upstream (the search runs under no_grad there too, and lbs_weights only enter a `>` gate)."""
    if points.ndim != 3 or template_points.ndim != 3 or lbs_weights.ndim != 2 or verts_transform.ndim != 4:
        raise ValueError("smpl_lbsmap_top_k: expected points [B,n,3], template_points [B,m,3], lbs_weights [m,J], verts_transform [B,m,4,4]")
    outs = [_LbsMapTopK.apply(lbs_weights, verts_transform[b], points[b], template_points[b], int(K),
                              addition_info[b] if addition_info is not None else None) for b in range(points.shape[0])]
    return tuple(torch.stack([o[k] for o in outs], 0) for k in range(len(outs[0])))


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
