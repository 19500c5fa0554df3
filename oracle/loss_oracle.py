"""CPU oracle for row f-5, the photometric loss (test infrastructure only -- never imported by the product path).

Restates l1_loss and ssim of /root/reference/hugs/losses/utils.py:54-58,65-108 in numpy float64, with the window built
the way the reference builds it (utils.py:65-74: fp32 Gaussian of 11 taps, sigma 1.5, normalised in fp32, outer product in
fp32) and applied as the full 11x11 zero-padded correlation per channel -- NOT separably: the kernel under test takes the
11 + 11 route, this checker the direct one.  The gradient is the analytic adjoint (three correlations of the per-pixel
partials), written independently of the kernel's formulas' arrangement.
Pinned by tests/golden/reference_substeps.npz `loss_*`: values and autograd gradients of the reference's own two functions,
imported from /root/reference by tests/golden/make_golden.py and run on CPU (fp32).
"""
from math import exp

import numpy as np
from scipy.signal import correlate2d

C1, C2 = np.float32(0.01 ** 2), np.float32(0.03 ** 2)


def window_1d():
    g = np.array([exp(-(x - 11 // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)], np.float32)   # utils.py:66
    # utils.py:67 -- torch's sum of these 11 floats rounds to 3.7592328 (what the exact sum rounds to; adding them one by one
    # in fp32 gives 3.7592325 and a window that is off by an ulp in nine taps): the window is pinned to the reference's by
    # tests/golden/reference_loss.npz `window_1d`
    return (g / np.float32(g.astype(np.float64).sum())).astype(np.float32)


def window_2d():
    w = window_1d()
    return (w[:, None] * w[None, :]).astype(np.float32)                                               # utils.py:72 (fp32 products)


def _corr(img, w):
    return np.stack([correlate2d(c, w, mode="same", boundary="fill", fillvalue=0.0) for c in img])


def _moments(x, y):
    w = window_2d().astype(np.float64)
    return _corr(x, w), _corr(y, w), _corr(x * x, w), _corr(y * y, w), _corr(x * y, w)


def ssim_map(img1, img2):
    """img1, img2 [C,H,W] -> the SSIM map [C,H,W] in float64 (utils.py:88-103)."""
    x, y = np.asarray(img1, np.float64), np.asarray(img2, np.float64)
    mu1, mu2, e11, e22, e12 = _moments(x, y)
    s1, s2, s12 = e11 - mu1 * mu1, e22 - mu2 * mu2, e12 - mu1 * mu2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))


def ssim(img1, img2):
    return float(ssim_map(img1, img2).mean())                                                         # utils.py:105-106


def l1_loss(network_output, gt, mask=None):
    d = np.abs(np.asarray(network_output, np.float64) - np.asarray(gt, np.float64))
    return float(d.sum() / np.asarray(mask, np.float64).sum()) if mask is not None else float(d.mean())   # utils.py:54-58


def grad(img1, img2, g_ssim_mean=0.0, g_l1_sum=0.0):
    """d(g_ssim_mean * ssim(img1, img2) + g_l1_sum * sum|img1 - img2|) / d img1, float64 [C,H,W]."""
    x, y = np.asarray(img1, np.float64), np.asarray(img2, np.float64)
    w = window_2d().astype(np.float64)
    mu1, mu2, e11, e22, e12 = _moments(x, y)
    A, B = 2 * mu1 * mu2 + C1, 2 * (e12 - mu1 * mu2) + C2
    Cc, D = mu1 * mu1 + mu2 * mu2 + C1, (e11 - mu1 * mu1) + (e22 - mu2 * mu2) + C2
    m = A * B / (Cc * D)
    # partials of the map w.r.t. the three img1-dependent moments (quotient rule; B may pass through 0 where the two
    # patches are anti-correlated, so nothing divides by A or B)
    d_mu1 = (2 * mu2 * B - 2 * mu2 * A) / (Cc * D) - m * (2 * mu1 / Cc - 2 * mu1 / D)
    d_e11 = -m / D
    d_e12 = 2 * A / (Cc * D)
    gm = g_ssim_mean / m.size
    out = gm * (_corr(d_mu1, w) + 2 * x * _corr(d_e11, w) + y * _corr(d_e12, w))   # the window is symmetric: adjoint = correlation
    return out + g_l1_sum * np.sign(x - y)
