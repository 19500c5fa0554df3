"""Learned-LBS skinning of the human Gaussians, fused (second half).

Drop-ins, with the reference's names, argument order and return values, for

    lbs_extra(A, v_shaped, posedirs, lbs_weights, pose, disable_posedirs=False, pose2rot=True)
                                                            /root/reference/hugs/models/modules/lbs.py:19-73
        -> (verts, A, T, v_posed, v_shaped), called every training step at hugs/models/hugs_trimlp.py:477-489

and `lbs_skin(A, weights, v, rotmat)`, the same skinning plus the rotation product that consumes T right after it
(`deformed_gs_rotmat = lbs_T[:, :3, :3] @ gs_rotmat`, hugs_trimlp.py:517) in one kernel.

The blend of the joint transforms, the point transform and the rotation product -- and all of their backward -- run in
hand-written HIP (csrc/lbs.hip) behind the C ABI; what stays in torch is what is a plain library op: the optional
pose-corrective matmul `pose_feature @ posedirs` (rocBLAS; every release config sets disable_posedirs: true) and
batch_rodrigues in front of it.  No CPU fallback.
"""
import ctypes as C

import torch

from diff_gaussian_rasterization import _load, _raise_last, _require_gpu

_bound = False


def _lib():
    global _bound
    lib = _load()
    if not _bound:
        p = C.c_void_p
        lib.hgs_lbs_skin_forward.restype = C.c_int32
        lib.hgs_lbs_skin_forward.argtypes = [C.c_int32, C.c_int32, p, p, p, p, p, p, p, p]
        lib.hgs_lbs_skin_backward_workspace.restype = C.c_size_t
        lib.hgs_lbs_skin_backward_workspace.argtypes = [C.c_int32, C.c_int32]
        lib.hgs_lbs_skin_backward.restype = C.c_int32
        lib.hgs_lbs_skin_backward.argtypes = [C.c_int32, C.c_int32] + [p] * 14
        _bound = True
    return lib


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


class _LbsSkin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, weights, v, rotmat):
        lib = _lib()
        _require_gpu(v, "v")
        A, weights, v = _f32c(A.reshape(-1, 16)), _f32c(weights), _f32c(v)
        rotmat = None if rotmat is None else _f32c(rotmat.reshape(-1, 9))
        n, J = v.shape[0], A.shape[0]
        if weights.shape != (n, J) or v.shape != (n, 3) or (rotmat is not None and rotmat.shape[0] != n):
            raise ValueError("lbs_skin: expected A [J,4,4], weights [n,J], v [n,3], rotmat [n,3,3]")
        dev = v.device
        T = torch.empty(n, 4, 4, dtype=torch.float32, device=dev)
        verts = torch.empty(n, 3, dtype=torch.float32, device=dev)
        rot = torch.empty(n, 3, 3, dtype=torch.float32, device=dev) if rotmat is not None else None
        with torch.cuda.device(dev):
            rc = lib.hgs_lbs_skin_forward(n, J, A.data_ptr(), weights.data_ptr(), v.data_ptr(), _ptr(rotmat), T.data_ptr(),
                                          verts.data_ptr(), _ptr(rot), torch.cuda.current_stream(dev).cuda_stream)
        if rc < 0:
            _raise_last(lib, "lbs_skin")
        ctx.save_for_backward(A, weights, v, T, *(() if rotmat is None else (rotmat,)))
        ctx.has_rot = rotmat is not None
        ctx.set_materialize_grads(False)
        if rot is None:
            return verts, T
        return verts, T, rot

    @staticmethod
    def backward(ctx, g_verts, g_T, g_rot=None):
        lib = _lib()
        A, weights, v, T = ctx.saved_tensors[:4]
        rotmat = ctx.saved_tensors[4] if ctx.has_rot else None
        n, J = v.shape[0], A.shape[0]
        dev = v.device
        g_verts = None if g_verts is None else _f32c(g_verts)
        g_T = None if g_T is None else _f32c(g_T)
        g_rot = None if g_rot is None else _f32c(g_rot)
        dA = torch.empty(J, 4, 4, dtype=torch.float32, device=dev)
        dW = torch.empty(n, J, dtype=torch.float32, device=dev)
        dv = torch.empty(n, 3, dtype=torch.float32, device=dev)
        dR = torch.empty(n, 3, 3, dtype=torch.float32, device=dev) if rotmat is not None else None
        ws = torch.empty(lib.hgs_lbs_skin_backward_workspace(n, J), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = lib.hgs_lbs_skin_backward(n, J, A.data_ptr(), weights.data_ptr(), v.data_ptr(), _ptr(rotmat), T.data_ptr(),
                                           _ptr(g_verts), _ptr(g_T), _ptr(g_rot), dA.data_ptr(), dW.data_ptr(), dv.data_ptr(),
                                           _ptr(dR), ws.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        if rc < 0:
            _raise_last(lib, "lbs_skin backward")
        return dA, dW, dv, dR


def lbs_skin(A, weights, v, rotmat=None):
    """A [J,4,4] joint transforms, weights [n,J], v [n,3] (v_posed), rotmat [n,3,3] or None
    -> (verts [n,3], T [n,4,4], T[:, :3,:3] @ rotmat or None); differentiable in all four inputs."""
    if rotmat is None:
        verts, T = _LbsSkin.apply(A, weights, v, None)
        return verts, T, None
    return _LbsSkin.apply(A, weights, v, rotmat)


def batch_rodrigues(rot_vecs, epsilon=1e-8):
    """smplx.lbs.batch_rodrigues (third-party, absent from /root/reference; published formula): axis-angle [N,3] ->
    [N,3,3].  Plain torch ops: it sits in front of a library matmul on a path every release config disables."""
    angle = torch.norm(rot_vecs + epsilon, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos, sin = torch.cos(angle)[:, None], torch.sin(angle)[:, None]
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros_like(rx)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(-1, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device)[None]
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def lbs_extra(A, v_shaped, posedirs, lbs_weights, pose, disable_posedirs=False, pose2rot=True):
    """lbs.py:19-73, same arguments and return tuple (verts, A, T, v_posed, v_shaped)."""
    batch_size = A.shape[0]
    if disable_posedirs:
        v_posed = v_shaped
    else:
        ident = torch.eye(3, dtype=A.dtype, device=A.device)
        if pose2rot:
            rot_mats = batch_rodrigues(pose.view(-1, 3)).view(batch_size, -1, 3, 3)
            pose_feature = (rot_mats[:, 1:, :, :] - ident).view(batch_size, -1)
        else:
            pose_feature = (pose[:, 1:].view(batch_size, -1, 3, 3) - ident).view(batch_size, -1)
        v_posed = torch.matmul(pose_feature, posedirs).view(batch_size, -1, 3) + v_shaped
    verts, Ts = [], []
    for b in range(batch_size):
        vb, Tb, _ = lbs_skin(A[b], lbs_weights, v_posed[b], None)
        verts.append(vb)
        Ts.append(Tb)
    return torch.stack(verts), A, torch.stack(Ts), v_posed, v_shaped
