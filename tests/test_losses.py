"""Row f-5 -- the photometric loss (l1_loss + ssim, /root/reference/hugs/losses/utils.py:54-108).
CPU: the numpy oracle (direct 11x11 correlation, float64) against values and autograd gradients produced by the reference's
own functions (tests/golden/make_golden_loss.py compiles them from /root/reference and runs them on CPU).
GPU: the fused HIP kernels (through the C ABI and the drop-in Python functions) against the oracle, the golden vectors and,
at 1080p, a torch restatement run on the same GPU.  Tolerances (fp32): values 2e-5; gradients 1e-4 of the largest (floor 1e-8)
gradient entry -- the kernel sums 11 + 11 taps in fp32 where conv2d sums 121 in its own order."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_oracle as lo

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_loss.npz"))
CASES = ["smooth", "noise", "same", "tiny", "ragged"]
VALUE_TOL, GRAD_TOL = 2e-5, 1e-4


def test_oracle_window_is_the_references():
    assert np.array_equal(lo.window_1d(), G["window_1d"])
    assert np.array_equal(lo.window_2d(), G["window_2d"])


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference_values_and_gradients(case):
    x, y = G[f"{case}_x"], G[f"{case}_y"]
    assert abs(lo.ssim(x, y) - float(G[f"{case}_ssim"])) <= VALUE_TOL
    assert abs(lo.l1_loss(x, y) - float(G[f"{case}_l1"])) <= VALUE_TOL * max(1.0, float(G[f"{case}_l1"]))
    for g_name, gs, gl in ((f"{case}_grad_ssim", 1.0, 0.0), (f"{case}_grad", -0.2, 0.8 / x.size)):
        ref = G[g_name].astype(np.float64)
        got = lo.grad(x, y, g_ssim_mean=gs, g_l1_sum=gl)
        assert np.abs(got - ref).max() <= GRAD_TOL * max(np.abs(ref).max(), 1e-4), g_name


def test_oracle_masked_l1_is_the_references():
    x, y, mask = G["smooth_x"], G["smooth_y"], G["masked_mask"]
    assert abs(lo.l1_loss(x, y, mask) - float(G["masked_l1"])) <= VALUE_TOL * float(G["masked_l1"])
    got = lo.grad(x, y, g_l1_sum=1.0 / mask.sum())
    assert np.abs(got - G["masked_grad"]).max() <= 1e-6 * np.abs(G["masked_grad"]).max()


def test_oracle_gradient_is_the_finite_difference_of_its_value():
    r = np.random.default_rng(3)
    x, y = r.random((2, 9, 13)), r.random((2, 9, 13))
    g = lo.grad(x, y, g_ssim_mean=1.0)
    for _ in range(12):
        c, i, j = r.integers(0, 2), r.integers(0, 9), r.integers(0, 13)
        e = np.zeros_like(x)
        e[c, i, j] = 1e-6
        fd = (lo.ssim(x + e, y) - lo.ssim(x - e, y)) / 2e-6
        assert abs(fd - g[c, i, j]) <= 1e-6 * max(1.0, abs(fd)) + 1e-9


# ---------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def device():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _fused(x, y, device, mask=None):
    from hugs_amd.losses import l1_loss, ssim
    tx = torch.from_numpy(x.copy()).to(device).requires_grad_(True)
    ty = torch.from_numpy(y).to(device)
    s = ssim(tx, ty)
    l1 = l1_loss(tx, ty, mask=None if mask is None else torch.from_numpy(mask).to(device))
    return tx, s, l1


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_matches_the_reference_vectors_and_the_oracle(case, device):
    x, y = G[f"{case}_x"], G[f"{case}_y"]
    tx, s, l1 = _fused(x, y, device)
    assert abs(s.item() - float(G[f"{case}_ssim"])) <= VALUE_TOL and abs(s.item() - lo.ssim(x, y)) <= VALUE_TOL
    assert abs(l1.item() - float(G[f"{case}_l1"])) <= VALUE_TOL * max(1.0, float(G[f"{case}_l1"]))
    (0.2 * (1.0 - s) + 0.8 * l1).backward()                          # hugs/losses/loss.py:96,107 with its default weights
    ref = G[f"{case}_grad"]
    got = tx.grad.cpu().numpy()
    assert np.abs(got - ref).max() <= GRAD_TOL * max(np.abs(ref).max(), 1e-4)
    assert np.abs(got - lo.grad(x, y, g_ssim_mean=-0.2, g_l1_sum=0.8 / x.size)).max() <= GRAD_TOL * max(np.abs(ref).max(), 1e-4)


@pytest.mark.gpu
def test_hip_ssim_alone_and_masked_l1(device):
    from hugs_amd.losses import l1_loss, ssim
    x, y, mask = G["smooth_x"], G["smooth_y"], G["masked_mask"]
    tx = torch.from_numpy(x.copy()).to(device).requires_grad_(True)
    ssim(tx, torch.from_numpy(y).to(device)).backward()
    assert np.abs(tx.grad.cpu().numpy() - G["smooth_grad_ssim"]).max() <= GRAD_TOL * np.abs(G["smooth_grad_ssim"]).max()
    tx2, _, lm = _fused(x, y, device, mask)
    assert abs(lm.item() - float(G["masked_l1"])) <= VALUE_TOL * float(G["masked_l1"])
    lm.backward()
    assert np.abs(tx2.grad.cpu().numpy() - G["masked_grad"]).max() <= 1e-6 * np.abs(G["masked_grad"]).max()
    with torch.no_grad():                                            # forward only: no maps are kept
        v = ssim(torch.from_numpy(x).to(device), torch.from_numpy(y).to(device))
    assert abs(v.item() - float(G["smooth_ssim"])) <= VALUE_TOL and not v.requires_grad


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 16, 64), (3, 17, 65), (1, 1, 1), (3, 130, 200), (2, 3, 33, 47)])
def test_hip_against_the_oracle_at_tile_edges_and_batches(shape, device):
    from hugs_amd.losses import l1_ssim
    r = np.random.default_rng(sum(shape))
    x, y = r.random(shape).astype(np.float32), r.random(shape).astype(np.float32)
    tx = torch.from_numpy(x.copy()).to(device).requires_grad_(True)
    l1, s = l1_ssim(tx, torch.from_numpy(y).to(device))
    (s + 3.0 * l1).backward()
    xs, ys = (x, y) if len(shape) == 4 else (x[None], y[None])
    want_s = np.mean([lo.ssim(a, b) for a, b in zip(xs, ys)])
    want_g = np.stack([lo.grad(a, b, g_ssim_mean=1.0 / len(xs), g_l1_sum=3.0 / x.size) for a, b in zip(xs, ys)]).reshape(shape)
    assert abs(s.item() - want_s) <= VALUE_TOL and abs(l1.item() - np.abs(x.astype(np.float64) - y).mean()) <= VALUE_TOL
    assert np.abs(tx.grad.cpu().numpy() - want_g).max() <= GRAD_TOL * np.abs(want_g).max()


def _torch_statements(x, y):
    """The same two quantities from plain torch ops (depthwise conv2d with the 11x11 window): the fp32 reference of the op."""
    w1 = torch.from_numpy(lo.window_1d()).to(device=x.device, dtype=x.dtype)
    win = (w1[:, None] * w1[None, :]).expand(x.shape[0], 1, 11, 11).contiguous()
    conv = lambda t: torch.nn.functional.conv2d(t[None], win, padding=5, groups=x.shape[0])[0]
    mu1, mu2 = conv(x), conv(y)
    s1, s2, s12 = conv(x * x) - mu1 * mu1, conv(y * y) - mu2 * mu2, conv(x * y) - mu1 * mu2
    m = ((2 * mu1 * mu2 + 0.01 ** 2) * (2 * s12 + 0.03 ** 2)) / ((mu1 * mu1 + mu2 * mu2 + 0.01 ** 2) * (s1 + s2 + 0.03 ** 2))
    return m.mean(), (x - y).abs().mean()


@pytest.mark.gpu
def test_hip_full_size_against_torch_on_the_same_gpu_and_properties(device):
    """1080p (BASELINE's image size): equal to the torch statements within the tolerances; an image against itself scores 1
    with a zero SSIM gradient; the result is bit-reproducible; ssim() then l1_loss() on one pair is one pass."""
    from hugs_amd import losses
    g = torch.Generator(device="cpu").manual_seed(0)
    y = torch.rand(3, 1080, 1920, generator=g).to(device)
    y = torch.nn.functional.avg_pool2d(y[None], 9, 1, 4)[0].contiguous()          # image-like: smooth
    x = (y + 0.03 * torch.randn(y.shape, generator=g).to(device)).clamp(0, 1).requires_grad_(True)
    s, l1 = losses.ssim(x, y), losses.l1_loss(x, y)
    assert s._base is l1._base and losses._LAST["entry"][3]() is s._base           # the second call reused the first pass
    (0.2 * (1.0 - s) + 0.8 * l1).backward()
    got = x.grad.clone()
    x.grad = None
    ts, tl1 = _torch_statements(x, y)
    (0.2 * (1.0 - ts) + 0.8 * tl1).backward()
    assert abs(s.item() - ts.item()) <= VALUE_TOL and abs(l1.item() - tl1.item()) <= VALUE_TOL * tl1.item()
    # (both sides fp32 here, conv2d's 121-term sums against 11 + 11: three times the tolerance used against the fp64 oracle)
    assert (got - x.grad).abs().max().item() <= 3 * GRAD_TOL * x.grad.abs().max().item()
    x.grad = None
    s2 = losses.ssim(x, y)
    (0.2 * (1.0 - s2) + 0.8 * losses.l1_loss(x, y)).backward()
    assert s2.item() == s.item() and torch.equal(x.grad, got)                      # fixed-order sums: bit-reproducible
    z = y.clone().requires_grad_(True)
    one = losses.ssim(z, y)
    one.backward()
    assert abs(one.item() - 1.0) <= 1e-6 and z.grad.abs().max().item() <= 1e-9


@pytest.mark.gpu
def test_hip_loss_errors(device):
    from hugs_amd import losses
    a = torch.rand(3, 8, 8, device=device)
    with pytest.raises(ValueError):
        losses.ssim(a, a[:, :4])
    with pytest.raises(NotImplementedError):
        losses.ssim(a, a.clone().requires_grad_(True))
    with pytest.raises(NotImplementedError):
        losses.ssim(a, a, window_size=7)
    with pytest.raises(RuntimeError):
        losses.l1_loss(a.double(), a.double())
    with pytest.raises(RuntimeError):
        losses.ssim(a.cpu(), a.cpu())


@pytest.mark.gpu
def test_shared_pass_is_found_again_only_while_the_caller_holds_the_result(device):
    """`ssim(a, b)` then `l1_loss(a, b)` share one pass over the images -- through weak references: while the first
    result is held the second call is the same node; once it is dropped, the partials go with it and the next call computes
    afresh; the backward of ANOTHER graph does not disturb a live entry."""
    import gc
    from hugs_amd import losses
    r = np.random.default_rng(0)
    a = torch.from_numpy(r.random((3, 40, 56)).astype(np.float32)).to(device).requires_grad_(True)
    b = torch.from_numpy(r.random((3, 40, 56)).astype(np.float32)).to(device)
    s = losses.ssim(a, b)
    l1 = losses.l1_loss(a, b)
    assert l1._base is s._base                                       # views of one result: one pass
    other = losses.ssim(a.detach().clone().requires_grad_(True), b)  # another graph; replaces the entry
    other.backward()
    s2 = losses.ssim(a, b)
    (s2 + l1).backward()                                             # both graphs still differentiate
    assert a.grad is not None and torch.isfinite(a.grad).all()
    s3 = losses.ssim(a, b)
    import weakref
    base, maps = weakref.ref(s3._base), weakref.ref(s3._base.grad_fn.saved_tensors[2])
    assert base() is not None and maps() is not None
    del s3
    gc.collect()
    assert base() is None and maps() is None                         # the dropped result kept nothing alive: no output, no partials
    s4 = losses.ssim(a, b)
    assert losses._LAST.get("entry") is None or losses._LAST["entry"][3]() is s4._base
