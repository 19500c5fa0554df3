"""Fused SceneGS.forward (row f-6): the statements that turn the scene model's raw parameters into the rasterizer's inputs on
every training step, /root/reference/hugs/models/scene.py:147-160 --

    def forward(self):                                   # in hugs/models/scene.py
        return hugs_amd.scene_forward.scene_forward(self._xyz, self._scaling, self._rotation, self._opacity,
                                                    self._features_dc, self._features_rest, self.active_sh_degree)

exp / normalize / sigmoid / cat in one HIP kernel, their backward in one more (the reference: 5 + ~17 torch kernels).  Same
dict, same keys, same values (fp32: expf and the division by the norm are correctly rounded here as there).  No CPU fallback.
"""
import ctypes as C

import torch

from diff_gaussian_rasterization import _load, _raise_last, _require_gpu, _stream_ptr

_PROTO = False


def _aligned(t):
    """contiguous AND 16-byte aligned: the row kernels move float4s.  A contiguous view whose storage offset is not a
    multiple of four floats -- a gradient that narrow / split / cat-backward carved out of a packed buffer -- is cloned
    (`.contiguous()` alone returns such a view unchanged and the library then refuses its pointer)."""
    if t is None:
        return None
    t = t.contiguous()
    return t.clone() if t.data_ptr() % 16 else t



def _lib():
    global _PROTO
    lib = _load()
    if not _PROTO:
        lib.hgs_scene_forward.restype = C.c_int32
        lib.hgs_scene_forward.argtypes = [C.c_int32, C.c_int32] + [C.c_void_p] * 10
        lib.hgs_scene_backward.restype = C.c_int32
        lib.hgs_scene_backward.argtypes = [C.c_int32, C.c_int32] + [C.c_void_p] * 13
        _PROTO = True
    return lib


class _SceneActivations(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scaling, rotation, opacity, features_dc, features_rest):
        lib = _lib()
        P, M = scaling.shape[0], 1 + features_rest.shape[1]
        dev = scaling.device
        scales, rotq = torch.empty_like(scaling), torch.empty_like(rotation)
        opac = torch.empty_like(opacity)
        shs = torch.empty(P, M, 3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.hgs_scene_forward(P, M, scaling.data_ptr(), rotation.data_ptr(), opacity.data_ptr(), features_dc.data_ptr(),
                                       features_rest.data_ptr() if M > 1 else None, scales.data_ptr(), rotq.data_ptr(),
                                       opac.data_ptr(), shs.data_ptr(), _stream_ptr(dev))
        if rc < 0:
            _raise_last(lib, "scene_forward")
        ctx.save_for_backward(rotation, scales, opac)
        ctx.M = M
        ctx.set_materialize_grads(False)
        return scales, rotq, opac, shs

    @staticmethod
    def backward(ctx, g_scales, g_rotq, g_opac, g_shs):
        rotation, scales, opac = ctx.saved_tensors
        lib = _lib()
        P, M, dev = scales.shape[0], ctx.M, scales.device
        c = _aligned
        g_scales, g_rotq, g_opac, g_shs = c(g_scales), c(g_rotq), c(g_opac), c(g_shs)
        new = lambda g, shape: torch.empty(shape, dtype=torch.float32, device=dev) if g is not None else None
        d_scaling, d_rot, d_op = new(g_scales, (P, 3)), new(g_rotq, (P, 4)), new(g_opac, (P, 1))
        d_dc, d_rest = new(g_shs, (P, 1, 3)), new(g_shs, (P, M - 1, 3))
        ptr = lambda t: None if t is None or t.numel() == 0 else t.data_ptr()
        with torch.cuda.device(dev):
            rc = lib.hgs_scene_backward(P, M, rotation.data_ptr(), scales.data_ptr(), opac.data_ptr(), ptr(g_scales), ptr(g_rotq),
                                        ptr(g_opac), ptr(g_shs), ptr(d_scaling), ptr(d_rot), ptr(d_op), ptr(d_dc), ptr(d_rest),
                                        _stream_ptr(dev))
        if rc < 0:
            _raise_last(lib, "scene_backward")
        return d_scaling, d_rot, d_op, d_dc, d_rest


def scene_activations(scaling, rotation, opacity, features_dc, features_rest):
    """-> (exp(scaling), normalize(rotation), sigmoid(opacity), cat(features_dc, features_rest, dim=1)), one kernel."""
    named = (("_scaling", scaling, 3), ("_rotation", rotation, 4), ("_opacity", opacity, 1))
    P = scaling.shape[0]
    for name, t, w in named:
        _require_gpu(t, name)
        if t.dtype != torch.float32 or t.shape != (P, w):
            raise RuntimeError(f"{name} must be a float32 tensor of shape [{P}, {w}]")
    for name, t in (("_features_dc", features_dc), ("_features_rest", features_rest)):
        _require_gpu(t, name)
        if t.dtype != torch.float32 or t.ndim != 3 or t.shape[0] != P or t.shape[2] != 3:
            raise RuntimeError(f"{name} must be a float32 tensor of shape [{P}, k, 3]")
    if features_dc.shape[1] != 1:
        raise RuntimeError("_features_dc must have one coefficient per Gaussian")
    return _SceneActivations.apply(_aligned(scaling), _aligned(rotation), _aligned(opacity), _aligned(features_dc), _aligned(features_rest))


def scene_forward(xyz, scaling, rotation, opacity, features_dc, features_rest, active_sh_degree):
    """The dict SceneGS.forward returns (scene.py:153-160)."""
    scales, rotq, opac, shs = scene_activations(scaling, rotation, opacity, features_dc, features_rest)
    return {"xyz": xyz, "scales": scales, "rotq": rotq, "shs": shs, "opacity": opac, "active_sh_degree": active_sh_degree}
