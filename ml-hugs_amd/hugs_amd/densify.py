"""Fused densification statistics.

Drop-in for the two statements that open `GaussianTrainer.scene_densification` / `human_densification`
(/root/reference/hugs/trainer/gs_trainer.py:406-411, 429-435) together with `add_densification_stats`
(/root/reference/hugs/models/scene.py:460-462, hugs/models/hugs_trimlp.py:880-882):

    model.max_radii2D[vis] = torch.max(model.max_radii2D[vis], radii[vis])
    model.xyz_gradient_accum[vis] += torch.norm(viewspace_points.grad[:vis.shape[0]][vis, :2], dim=-1, keepdim=True)
    model.denom[vis] += 1

One HIP kernel instead of four boolean-indexed torch ops; same in-place semantics, including the reference's
habit of pairing the FIRST n rows of the gradient with the model's n Gaussians.  No CPU fallback.
"""
import ctypes as C

import torch

from diff_gaussian_rasterization import _load, _raise_last, _require_gpu, _stream_ptr


def update_densification_stats(max_radii2D, xyz_gradient_accum, denom, viewspace_point_tensor, visibility_filter, radii):
    """In-place update of the three per-Gaussian statistics tensors (fp32; [n], [n,1], [n,1])."""
    lib = _load()
    grad = viewspace_point_tensor.grad if viewspace_point_tensor.grad is not None else None
    if grad is None:
        raise RuntimeError("viewspace_point_tensor has no .grad (call backward first)")
    n = int(visibility_filter.shape[0])
    for t, name in ((max_radii2D, "max_radii2D"), (xyz_gradient_accum, "xyz_gradient_accum"), (denom, "denom"), (grad, "grad")):
        _require_gpu(t, name)
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError(f"{name} must be a contiguous float32 tensor")
    if grad.shape[0] < n or max_radii2D.numel() != n or xyz_gradient_accum.numel() != n or denom.numel() != n:
        raise RuntimeError("size mismatch between the statistics tensors, the filter and the gradient")
    vis = visibility_filter.contiguous().view(torch.uint8) if visibility_filter.dtype == torch.bool else \
        (visibility_filter != 0).contiguous().view(torch.uint8)
    rad = radii.to(torch.int32).contiguous()
    lib.hgs_densification_stats.restype = C.c_int32
    lib.hgs_densification_stats.argtypes = [C.c_int32] + [C.c_void_p] * 7
    with torch.cuda.device(grad.device):
        rc = lib.hgs_densification_stats(n, grad.data_ptr(), rad.data_ptr(), vis.data_ptr(), max_radii2D.data_ptr(),
                                         xyz_gradient_accum.data_ptr(), denom.data_ptr(), _stream_ptr(grad.device))
    if rc < 0:
        _raise_last(lib, "densification_stats")
