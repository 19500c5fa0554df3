"""Fused photometric loss (row f-5): drop-ins for `l1_loss` and `ssim` of /root/reference/hugs/losses/utils.py:54-58,77-108,
the two terms every training step computes on the rendered image (hugs/losses/loss.py:88-107) and again on the human-only
render (:128-137).

    from hugs_amd.losses import l1_loss, ssim            # instead of `from .utils import l1_loss, ssim` (loss.py:13)

Same signatures, same values (fp32; the 11x11 window is applied as 11 + 11 taps, so sums differ from conv2d's in the last
bits), same gradient with respect to the first argument -- the render; the target never requires a gradient at the
reference's call sites and asking for one raises.  `l1_ssim(pred, gt)` returns both terms from ONE pass over the images
(`ssim` and `l1_loss` called one after the other on the same pair share that pass too: the second call finds the first
one's result).  No CPU fallback: the HIP library does the work or the call raises.
"""
import ctypes as C
import os
import weakref

import torch

from diff_gaussian_rasterization import _load, _raise_last, _require_gpu, _stream_ptr

_PROTO = False


def _lib():
    global _PROTO
    lib = _load()
    if not _PROTO:
        lib.hgs_ssim_l1_workspace.restype = C.c_size_t
        lib.hgs_ssim_l1_workspace.argtypes = [C.c_int32] * 3
        lib.hgs_ssim_l1_forward.restype = C.c_int32
        lib.hgs_ssim_l1_forward.argtypes = [C.c_int32] * 3 + [C.c_void_p] * 6
        lib.hgs_ssim_l1_backward.restype = C.c_int32
        lib.hgs_ssim_l1_backward.argtypes = [C.c_int32] * 3 + [C.c_void_p] * 7
        _PROTO = True
    return lib


class _SsimL1(torch.autograd.Function):
    """(pred [C,H,W], gt [C,H,W]) -> ONE tensor [ssim mean, l1 mean, l1 sum]: hgs_ssim_l1_forward / hgs_ssim_l1_backward.
    (One output: the terms the callers hand out are views of it, and a view keeps its base alive -- which is what lets the
    shared-pass entry below hold a WEAK reference and still be found while any of the terms is in use.)"""

    @staticmethod
    def forward(ctx, pred, gt):
        lib = _lib()
        Cn, H, W = pred.shape
        need_grad = pred.requires_grad
        out = torch.empty(3, dtype=torch.float32, device=pred.device)
        maps = torch.empty(3, Cn, H, W, dtype=torch.float32, device=pred.device) if need_grad else None
        ws = torch.empty(lib.hgs_ssim_l1_workspace(Cn, H, W), dtype=torch.uint8, device=pred.device)
        with torch.cuda.device(pred.device):
            rc = lib.hgs_ssim_l1_forward(Cn, H, W, pred.data_ptr(), gt.data_ptr(), maps.data_ptr() if need_grad else None,
                                         ws.data_ptr(), out.data_ptr(), _stream_ptr(pred.device))
        if rc < 0:
            _raise_last(lib, "ssim_l1_forward")
        ctx.save_for_backward(pred, gt, maps)
        return out

    @staticmethod
    def backward(ctx, g):
        hit = _LAST.get("entry")                  # (THIS graph is spent: a later call on the same pair computes afresh;
        if hit is not None and hit[4] == id(ctx):  #  an entry that belongs to another graph stays)
            _LAST.pop("entry", None)
        pred, gt, maps = ctx.saved_tensors
        lib = _lib()
        Cn, H, W = pred.shape
        # d(l1 mean) = d(l1 sum) / (C H W): one device scalar for the kernel, no host round trip
        g = g.to(torch.float32)
        g_s = g[0:1].contiguous()
        g_l1 = (g[1:2] / float(Cn * H * W) + g[2:3]).contiguous()
        grad = torch.empty_like(pred)
        with torch.cuda.device(pred.device):
            rc = lib.hgs_ssim_l1_backward(Cn, H, W, pred.data_ptr(), gt.data_ptr(), maps.data_ptr() if maps is not None else None,
                                          g_s.data_ptr() if g_s is not None else None, g_l1.data_ptr() if g_l1 is not None else None,
                                          grad.data_ptr(), _stream_ptr(pred.device))
        if rc < 0:
            _raise_last(lib, "ssim_l1_backward")
        return grad, None


def _prep(pred, gt):
    if pred.shape != gt.shape or pred.ndim not in (3, 4):
        raise ValueError("expected two images of the same shape, [C,H,W] or [B,C,H,W]")
    if gt.requires_grad:
        raise NotImplementedError("the fused loss differentiates with respect to its first argument only (the render)")
    _require_gpu(pred, "network_output")
    _require_gpu(gt, "gt")
    if pred.dtype != torch.float32 or gt.dtype != torch.float32:
        raise RuntimeError("the fused loss takes float32 images")
    return pred.contiguous(), gt.contiguous()


# ssim(a, b) followed by l1_loss(a, b) on the same tensors (loss.py:88-99) is one pass: the pair's result is found again
# while both tensors are alive and unchanged (same objects, same version counters, same storage), while the CALLER still
# holds the first call's result -- the entry keeps only weak references to the three outputs, so a result nobody uses takes
# its autograd node and the 3 x C x H x W partials (75 MB at 1080p) with it at once, not at the next call -- and until that
# graph's backward has run (a backward of ANOTHER graph leaves the entry alone).  `l1_ssim` is the explicit form of the same.
# HGS_LOSS_SHARE_PASS=0 turns the sharing off (every call computes afresh) -- for callers whose custom kernels rewrite a
# tensor's memory behind autograd's back, which no version counter records.
_LAST = {}
_SHARE = os.environ.get("HGS_LOSS_SHARE_PASS", "1") != "0"


def _terms(pred, gt):
    """[ssim mean, l1 mean, l1 sum] (one tensor) of one [C,H,W] pair, computed once per (pred, gt) pair and version."""
    key = (id(pred), id(gt), pred._version, gt._version, pred.data_ptr(), gt.data_ptr(), pred.requires_grad and torch.is_grad_enabled())
    hit = _LAST.get("entry")
    if _SHARE and hit is not None and hit[0] == key and hit[1]() is pred and hit[2]() is gt:
        out = hit[3]()
        if out is not None:
            return out
    out = _SsimL1.apply(pred, gt)
    if not _SHARE:
        return out
    try:
        _LAST["entry"] = (key, weakref.ref(pred), weakref.ref(gt), weakref.ref(out), id(out.grad_fn))
    except TypeError:
        _LAST.pop("entry", None)
    return out


def l1_ssim(network_output, gt):
    """-> (l1_loss(network_output, gt), ssim(network_output, gt)) from one pass over the two images."""
    pred, tgt = _prep(network_output, gt)
    if pred.ndim == 4:
        terms = [_terms(pred[i], tgt[i]) for i in range(pred.shape[0])]
        return torch.stack([t[1] for t in terms]).mean(), torch.stack([t[0] for t in terms]).mean()
    s, l1_mean, _ = _terms(pred, tgt)
    return l1_mean, s


def l1_loss(network_output, gt, mask=None):
    """utils.py:54-58: mean |a - b|, or sum |a - b| / mask.sum() with a mask."""
    pred, tgt = _prep(network_output, gt)
    if pred.ndim == 4:
        terms = [_terms(pred[i], tgt[i]) for i in range(pred.shape[0])]
        total = torch.stack([t[2] for t in terms]).sum()
        return total / mask.sum() if mask is not None else total / float(pred.numel())
    _, l1_mean, l1_sum = _terms(pred, tgt)
    return l1_sum / mask.sum() if mask is not None else l1_mean


def ssim(img1, img2, window_size=11, size_average=True, mask=None):
    """utils.py:77-108 (the `mask` argument is accepted and ignored, as there)."""
    if window_size != 11:
        raise NotImplementedError("ssim (MI355X): window_size 11 (all the reference uses)")
    pred, tgt = _prep(img1, img2)
    if pred.ndim == 3:
        return _terms(pred, tgt)[0]              # (size_average=False on a [C,H,W] image fails in the reference: mean(1) x3)
    per_image = torch.stack([_terms(pred[i], tgt[i])[0] for i in range(pred.shape[0])])
    return per_image.mean() if size_average else per_image
