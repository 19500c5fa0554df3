"""Row f-7 -- rotation_6d_to_matrix and matrix_to_quaternion (/root/reference/hugs/utils/rotations.py:552-573,94-156).
CPU: the numpy oracle against outputs and autograd gradients of the reference's own functions (tests/golden/make_golden_rotations.py).
GPU: the HIP kernels through the drop-in Python functions against the golden vectors and the oracle.  Tolerances (fp32):
values 2e-6 absolute (entries are O(1)); gradients 2e-5 of the largest entry of the tensor."""
import os

import numpy as np
import pytest
import torch

from oracle import rotation_oracle as ro

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_rotations.npz"))
VALUE_TOL, GRAD_TOL = 2e-6, 2e-5


def _grad_close(got, want):
    want = np.asarray(want, np.float64)
    finite = np.isfinite(want)
    return np.array_equal(np.isfinite(got), finite) and np.abs(np.asarray(got, np.float64)[finite] - want[finite]).max() <= GRAD_TOL * np.abs(want[finite]).max()


def test_oracle_matches_the_reference():
    ok = slice(2, None)      # rows 0, 1: a zero and two parallel vectors -- normalize's eps branch / a direction made of rounding noise
    assert np.abs(ro.rotation_6d_to_matrix(G["d6"])[ok] - G["d6_matrix"][ok]).max() <= VALUE_TOL
    assert np.abs(ro.rotation_6d_to_matrix(G["d6"])[0] - G["d6_matrix"][0]).max() <= VALUE_TOL    # zero first vector: zeros in rows 1 and 3
    assert np.abs(ro.matrix_to_quaternion(G["matrix"]) - G["quat"]).max() <= VALUE_TOL
    assert _grad_close(ro.rotation_6d_to_matrix_backward(G["d6"], G["d6_g"])[ok], G["d6_grad"][ok])
    assert _grad_close(ro.matrix_to_quaternion_backward(G["matrix"], G["quat_g"]), G["matrix_grad"])


@pytest.fixture(scope="module")
def device():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_hip_matches_the_reference_vectors(device):
    from hugs_amd.rotations import matrix_to_quaternion, rotation_6d_to_matrix
    t6 = torch.from_numpy(G["d6"].copy()).to(device).requires_grad_(True)
    R = rotation_6d_to_matrix(t6)
    assert R.shape == (300, 3, 3) and np.abs(R.detach().cpu().numpy()[2:] - G["d6_matrix"][2:]).max() <= VALUE_TOL
    assert np.abs(R.detach().cpu().numpy()[0] - G["d6_matrix"][0]).max() <= VALUE_TOL and torch.isfinite(R).all()
    R.backward(torch.from_numpy(G["d6_g"]).to(device))
    assert _grad_close(t6.grad.cpu().numpy()[2:], G["d6_grad"][2:])
    tM = torch.from_numpy(G["matrix"].copy()).to(device).requires_grad_(True)
    q = matrix_to_quaternion(tM)
    assert q.shape == G["quat"].shape and np.abs(q.detach().cpu().numpy() - G["quat"]).max() <= VALUE_TOL
    q.backward(torch.from_numpy(G["quat_g"]).to(device))
    assert _grad_close(tM.grad.cpu().numpy(), G["matrix_grad"])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1,), (257,), (110_210,), (2, 3, 5)])
def test_hip_against_the_oracle_with_leading_dimensions(shape, device):
    """HUGS's 110 210 human Gaussians; batched leading dimensions; the chain 6-D -> matrix -> quaternion of hugs_trimlp.py:418-419."""
    from hugs_amd.rotations import matrix_to_quaternion, rotation_6d_to_matrix
    r = np.random.default_rng(sum(shape))
    d6 = r.standard_normal(shape + (6,)).astype(np.float32)
    t6 = torch.from_numpy(d6.copy()).to(device).requires_grad_(True)
    R = rotation_6d_to_matrix(t6)
    q = matrix_to_quaternion(R)
    assert R.shape == shape + (3, 3) and q.shape == shape + (4,)
    wantR = ro.rotation_6d_to_matrix(d6)
    # Gram-Schmidt cancels |a2| down to |u|: the rounding error of b2 grows with that ratio.  Values: the tolerance scaled by
    # it; gradients: compared on the samples where it is moderate (the others: finite)
    a1, a2 = d6[..., :3].astype(np.float64), d6[..., 3:].astype(np.float64)
    b1 = a1 / np.linalg.norm(a1, axis=-1, keepdims=True)
    cond = np.linalg.norm(a2, axis=-1) / np.linalg.norm(a2 - (b1 * a2).sum(-1, keepdims=True) * b1, axis=-1)
    assert (np.abs(R.detach().cpu().numpy() - wantR).reshape(shape + (9,)).max(-1) <= VALUE_TOL * np.maximum(cond, 1.0)).all()
    Rn = R.detach().cpu().numpy()
    assert np.abs(q.detach().cpu().numpy() - ro.matrix_to_quaternion(Rn)).max() <= VALUE_TOL
    gq = r.standard_normal(shape + (4,)).astype(np.float32)
    q.backward(torch.from_numpy(gq).to(device))
    gR = ro.matrix_to_quaternion_backward(Rn, gq)
    want6, got6 = ro.rotation_6d_to_matrix_backward(d6, gR), t6.grad.cpu().numpy()
    tame = cond < 10.0
    assert tame.mean() > 0.85 and np.isfinite(got6).all()
    assert np.abs(got6[tame] - want6[tame]).max() <= 10 * GRAD_TOL * np.abs(want6[tame]).max()
    # unit quaternions of proper rotations, up to sign the same rotation
    qn = q.detach().cpu().numpy().reshape(-1, 4)
    assert np.abs(np.linalg.norm(qn, axis=1) - 1).max() <= 1e-5


@pytest.mark.gpu
def test_hip_rotation_errors(device):
    from hugs_amd.rotations import matrix_to_quaternion, rotation_6d_to_matrix
    with pytest.raises(ValueError):
        matrix_to_quaternion(torch.zeros(4, 3, 4, device=device))
    with pytest.raises(ValueError):
        rotation_6d_to_matrix(torch.zeros(4, 5, device=device))
    with pytest.raises(RuntimeError):
        matrix_to_quaternion(torch.zeros(4, 3, 3))
    with pytest.raises(RuntimeError):
        rotation_6d_to_matrix(torch.zeros(4, 6, device=device, dtype=torch.float64))
    assert matrix_to_quaternion(torch.zeros(0, 3, 3, device=device)).shape == (0, 4)


@pytest.mark.gpu
def test_hip_rotations_take_gradients_and_inputs_at_odd_storage_offsets(device):
    """a contiguous view whose storage offset is not a multiple of four floats -- what narrow / split / cat-backward
    hand out of a packed buffer -- used to make the row kernels refuse the pointer ('not 16-byte aligned') in backward.  The
    wrappers now realign such tensors: same values and gradients as with aligned ones."""
    from hugs_amd.rotations import matrix_to_quaternion, rotation_6d_to_matrix
    n = 1000
    r = np.random.default_rng(3)
    d6 = torch.from_numpy(r.standard_normal((n, 6)).astype(np.float32)).to(device)
    gq = torch.from_numpy(r.standard_normal((n, 4)).astype(np.float32)).to(device)
    out = {}
    for odd in (False, True):
        if odd:   # the same numbers, one float into a packed buffer: contiguous, data_ptr() % 16 == 4
            src6 = torch.cat([torch.zeros(1, device=device), d6.reshape(-1)])[1:].view(n, 6)
            g = torch.cat([torch.zeros(1, device=device), gq.reshape(-1)])[1:].view(n, 4)
            assert src6.is_contiguous() and src6.data_ptr() % 16 != 0 and g.data_ptr() % 16 != 0
        else:
            src6, g = d6.clone(), gq.clone()
        t6 = src6.detach().requires_grad_(True)
        q = matrix_to_quaternion(rotation_6d_to_matrix(t6))
        q.backward(g)
        out[odd] = (q.detach().clone(), t6.grad.clone())
    assert torch.equal(out[False][0], out[True][0]) and torch.equal(out[False][1], out[True][1])
