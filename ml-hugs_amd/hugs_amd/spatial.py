"""Storage order of the Gaussians.  The rasterizer's results do not depend on it (up to the summation order of float
atomics and the index tie-break of exactly equal depths).  Its binning kernels work on groups of ~800 Gaussians, and a
group whose members are neighbours on screen touches few tiles -- long runs per tile instead of one or two entries, few
returning atomics, coalesced key stores ("Binning").  Large frames (≥ 32 768 Gaussians, ≥ 4 096
tiles) form such groups themselves, with a counting sort by screen cell; smaller ones bin in STORAGE order, and a model
whose Gaussians are appended in arbitrary order (densification does that: /root/reference/hugs/models/scene.py:333-361)
can be put in Morton order now and then; `permute_model` applies one permutation to every per-Gaussian tensor of a dict."""
import numpy as np
import torch


def _spread3(v):
    """10 bits -> every third bit of 30"""
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


def morton_order(means3D):
    """Permutation (int64) that sorts [P,3] positions (numpy array or torch tensor, any device) by their 30-bit Morton
    code inside the positions' bounding box; non-finite positions go last."""
    if isinstance(means3D, torch.Tensor):
        if means3D.shape[0] == 0:
            return torch.zeros(0, dtype=torch.int64, device=means3D.device)
        x = means3D.detach().float()
        ok = torch.isfinite(x).all(dim=1)
        lo = torch.where(ok[:, None], x, torch.full_like(x, float("inf"))).amin(dim=0)
        hi = torch.where(ok[:, None], x, torch.full_like(x, float("-inf"))).amax(dim=0)
        q = ((x - lo) / (hi - lo).clamp_min(1e-30) * 1023.0).clamp(0, 1023).nan_to_num(0).to(torch.int64)
        code = _spread3(q[:, 0]) | (_spread3(q[:, 1]) << 1) | (_spread3(q[:, 2]) << 2)
        code = torch.where(ok, code, torch.full_like(code, 1 << 40))
        return torch.argsort(code, stable=True)
    x = np.asarray(means3D, np.float64)
    ok = np.isfinite(x).all(axis=1)
    lo, hi = (x[ok].min(axis=0), x[ok].max(axis=0)) if ok.any() else (np.zeros(3), np.ones(3))
    q = np.clip(np.nan_to_num((x - lo) / np.maximum(hi - lo, 1e-30) * 1023.0), 0, 1023).astype(np.int64)
    code = _spread3(q[:, 0]) | (_spread3(q[:, 1]) << 1) | (_spread3(q[:, 2]) << 2)
    code[~ok] = 1 << 40
    return np.argsort(code, kind="stable")


def permute_model(tensors, order):
    """{name: tensor}: every tensor whose first dimension is len(order) re-indexed by `order`, the others untouched."""
    n = len(order)
    return {k: (v[order] if hasattr(v, "shape") and len(v.shape) and v.shape[0] == n else v) for k, v in tensors.items()}
