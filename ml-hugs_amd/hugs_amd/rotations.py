"""Fused rotation conversions (row f-7): drop-ins for `rotation_6d_to_matrix` and `matrix_to_quaternion` of
/root/reference/hugs/utils/rotations.py:552-573,94-156 -- the statements that turn the human model's 6-D rotation output into
the quaternions the rasterizer takes, on every training step (hugs_trimlp.py:418-419,518):

    from hugs_amd.rotations import matrix_to_quaternion, rotation_6d_to_matrix     # instead of hugs.utils.rotations (hugs_trimlp.py:20-27)

Same signatures, any leading dimensions, same values and the same autograd gradients (the selected candidate only, zero
subgradient where a trace term is not positive, floor 0.1) -- one kernel per direction instead of ~15 / ~40 torch kernels, and
none of the two boolean-mask indexing steps that make the reference wait for the device twice per call.  No CPU fallback.
"""
import ctypes as C

import torch

from diff_gaussian_rasterization import _load, _raise_last, _require_gpu, _stream_ptr


def _aligned(t):
    """contiguous AND 16-byte aligned: the row kernels move float4s.  A contiguous view whose storage offset is not a
    multiple of four floats -- a gradient that narrow / split / cat-backward carved out of a packed buffer -- is cloned
    (`.contiguous()` alone returns such a view unchanged and the library then refuses its pointer)."""
    if t is None:
        return None
    t = t.contiguous()
    return t.clone() if t.data_ptr() % 16 else t



def _call(name, n, dev, *tensors):
    lib = _load()
    fn = getattr(lib, name)
    fn.restype = C.c_int32
    fn.argtypes = [C.c_int32] + [C.c_void_p] * (len(tensors) + 1)
    with torch.cuda.device(dev):
        rc = fn(n, *[t.data_ptr() for t in tensors], _stream_ptr(dev))
    if rc < 0:
        _raise_last(lib, name[4:])


class _MatrixToQuaternion(torch.autograd.Function):
    @staticmethod
    def forward(ctx, matrix):                                     # [n, 9] contiguous
        quat = torch.empty(matrix.shape[0], 4, dtype=torch.float32, device=matrix.device)
        _call("hgs_matrix_to_quaternion", matrix.shape[0], matrix.device, matrix, quat)
        ctx.save_for_backward(matrix)
        return quat

    @staticmethod
    def backward(ctx, g):
        (matrix,) = ctx.saved_tensors
        grad = torch.empty_like(matrix)
        _call("hgs_matrix_to_quaternion_backward", matrix.shape[0], matrix.device, matrix, _aligned(g), grad)
        return grad


class _Rotation6dToMatrix(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d6):                                         # [n, 6] contiguous
        matrix = torch.empty(d6.shape[0], 9, dtype=torch.float32, device=d6.device)
        _call("hgs_rotation_6d_to_matrix", d6.shape[0], d6.device, d6, matrix)
        ctx.save_for_backward(d6)
        return matrix

    @staticmethod
    def backward(ctx, g):
        (d6,) = ctx.saved_tensors
        grad = torch.empty_like(d6)
        _call("hgs_rotation_6d_to_matrix_backward", d6.shape[0], d6.device, d6, _aligned(g), grad)
        return grad


def _check(t, name):
    _require_gpu(t, name)
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32")


def matrix_to_quaternion(matrix):
    """(..., 3, 3) rotation matrices -> (..., 4) quaternions, real part first (rotations.py:105-156)."""
    if matrix.size(-1) != 3 or matrix.size(-2) != 3:
        raise ValueError(f"Invalid rotation matrix shape {matrix.shape}.")
    _check(matrix, "matrix")
    batch = matrix.shape[:-2]
    return _MatrixToQuaternion.apply(_aligned(matrix.reshape(-1, 9))).reshape(batch + (4,))


def rotation_6d_to_matrix(d6):
    """(*, 6) -> (*, 3, 3) by Gram-Schmidt (rotations.py:552-573)."""
    if d6.size(-1) != 6:
        raise ValueError(f"Invalid 6-D rotation shape {d6.shape}.")
    _check(d6, "d6")
    batch = d6.shape[:-1]
    return _Rotation6dToMatrix.apply(_aligned(d6.reshape(-1, 6))).reshape(batch + (3, 3))
