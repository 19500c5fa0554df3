"""SURVEY.md 8f row f-2, second half -- the learned-LBS skinning step of the human model (lbs_extra,
/root/reference/hugs/models/modules/lbs.py:19-73, called every training step at hugs_trimlp.py:477-489) and the rotation
product that follows it (hugs_trimlp.py:517).
CPU: the numpy oracle against vectors produced by the reference's own statements (forward and autograd backward).
GPU: the fused HIP kernels (through the C ABI and the drop-in Python functions) against oracle and golden vectors."""
import os

import numpy as np
import pytest
import torch

from oracle import lbs_oracle as lo

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_substeps.npz"))
TOL = dict(rtol=2e-5, atol=2e-6)     # fp32 sums of 24 (forward) / up to a few hundred (backward) products, order differs


def test_oracle_matches_reference_lbs_extra():
    for tag, disable in (("", True), ("_posedirs", False)):
        verts, _, T, v_posed, _ = lo.lbs_extra(G["lbs_A"][None], G["lbs_v"][None], G["lbs_posedirs"], G["lbs_weights"], G["lbs_pose"],
                                               disable_posedirs=disable)
        np.testing.assert_allclose(v_posed[0], G[f"lbs{tag}_v_posed"], **TOL)
        np.testing.assert_allclose(T[0], G[f"lbs{tag}_T"], **TOL)
        np.testing.assert_allclose(verts[0], G[f"lbs{tag}_verts"], **TOL)
    verts, T, rot = lo.skin(G["lbs_A"], G["lbs_weights"], G["lbs_v"], G["lbs_rotmat"])
    np.testing.assert_allclose(rot, G["lbs_rot"], **TOL)
    assert np.array_equal(verts, lo.lbs_extra(G["lbs_A"][None], G["lbs_v"][None], None, G["lbs_weights"], None, disable_posedirs=True)[0][0])


def test_oracle_backward_matches_reference_autograd():
    g = lo.skin_backward(G["lbs_A"], G["lbs_weights"], G["lbs_v"], G["lbs_rotmat"], G["lbs_g_verts"], G["lbs_g_T"], G["lbs_g_rot"])
    for k, ref in (("A", "lbs_dA"), ("weights", "lbs_dW"), ("v", "lbs_dv"), ("rotmat", "lbs_dR")):
        scale = np.abs(G[ref]).max()
        assert np.abs(g[k] - G[ref]).max() <= 3e-5 * scale, k
    # with posedirs the point gradient is unchanged (v_posed = v_shaped + offsets) -- what the drop-in relies on
    np.testing.assert_allclose(G["lbs_posedirs_dv"], lo.skin_backward(G["lbs_A"], G["lbs_weights"], G["lbs_posedirs_v_posed"], G["lbs_rotmat"],
                                                                      G["lbs_g_verts"], G["lbs_g_T"], G["lbs_g_rot"])["v"], rtol=3e-5, atol=3e-5)


def test_batch_rodrigues_is_a_rotation_and_matches_the_axis_angle_definition():
    r = np.random.default_rng(0).standard_normal((50, 3)).astype(np.float64)
    R = lo.batch_rodrigues(r, np.float64)
    # (the published formula takes the angle of r + 1e-8: the axis is off unit length by ~1e-8)
    np.testing.assert_allclose(R @ np.transpose(R, (0, 2, 1)), np.tile(np.eye(3), (50, 1, 1)), atol=1e-7)
    np.testing.assert_allclose(np.linalg.det(R), 1.0, atol=1e-7)
    np.testing.assert_allclose(np.einsum("nij,nj->ni", R, r), r, atol=1e-7)                   # the axis is fixed
    np.testing.assert_allclose(np.trace(R, axis1=1, axis2=2), 1 + 2 * np.cos(np.linalg.norm(r, axis=1)), atol=1e-7)


def _body(n, J, seed):
    r = np.random.default_rng(seed)
    A = np.tile(np.eye(4, dtype=np.float32), (J, 1, 1))
    A[:, :3, :] += 0.3 * r.standard_normal((J, 3, 4)).astype(np.float32)
    logit = 4.0 * r.standard_normal((n, J))
    W = (np.exp(logit) / np.exp(logit).sum(1, keepdims=True)).astype(np.float32)
    v = (r.standard_normal((n, 3)) * np.array([0.25, 0.6, 0.15])).astype(np.float32)
    R = lo.batch_rodrigues(r.standard_normal((n, 3)).astype(np.float32))
    return A, W, v, R, r


@pytest.mark.gpu
def test_hip_matches_golden_vectors_forward_and_backward(device):
    from hugs_amd.lbs import lbs_extra, lbs_skin
    t = lambda k, grad=False: torch.from_numpy(G[k].copy()).to(device).requires_grad_(grad)
    A, W, v, R = t("lbs_A", True), t("lbs_weights", True), t("lbs_v", True), t("lbs_rotmat", True)
    verts, T, rot = lbs_skin(A, W, v, R)
    np.testing.assert_allclose(verts.detach().cpu().numpy(), G["lbs_verts"], **TOL)
    np.testing.assert_allclose(T.detach().cpu().numpy(), G["lbs_T"], **TOL)
    np.testing.assert_allclose(rot.detach().cpu().numpy(), G["lbs_rot"], **TOL)
    ((verts * t("lbs_g_verts")).sum() + (T * t("lbs_g_T")).sum() + (rot * t("lbs_g_rot")).sum()).backward()
    for x, ref in ((A, "lbs_dA"), (W, "lbs_dW"), (v, "lbs_dv"), (R, "lbs_dR")):
        assert np.abs(x.grad.cpu().numpy() - G[ref]).max() <= 3e-5 * np.abs(G[ref]).max(), ref
    # the drop-in lbs_extra, both posedirs settings (the posedirs product stays a torch matmul in front of the kernel)
    for tag, disable in (("", True), ("_posedirs", False)):
        A2, W2, v2 = t("lbs_A", True), t("lbs_weights", True), t("lbs_v", True)
        verts, A_out, T, v_posed, v_shaped = lbs_extra(A2[None], v2[None], t("lbs_posedirs"), W2, t("lbs_pose"), disable_posedirs=disable)
        assert verts.shape == (1, 96, 3) and T.shape == (1, 96, 4, 4) and A_out is not None and v_shaped.shape == (1, 96, 3)
        np.testing.assert_allclose(verts[0].detach().cpu().numpy(), G[f"lbs{tag}_verts"], **TOL)
        np.testing.assert_allclose(v_posed[0].detach().cpu().numpy(), G[f"lbs{tag}_v_posed"], **TOL)
        rot = T[0][:, :3, :3] @ t("lbs_rotmat")
        ((verts[0] * t("lbs_g_verts")).sum() + (T[0] * t("lbs_g_T")).sum() + (rot * t("lbs_g_rot")).sum()).backward()
        for x, ref in ((A2, f"lbs{tag}_dA"), (W2, f"lbs{tag}_dW"), (v2, f"lbs{tag}_dv")):
            assert np.abs(x.grad.cpu().numpy() - G[ref]).max() <= 3e-5 * np.abs(G[ref]).max(), ref


@pytest.mark.gpu
@pytest.mark.parametrize("n,J", [(1, 24), (63, 24), (6890, 24), (110_210, 24), (1000, 7), (257, 32)])
def test_hip_skinning_against_the_oracle(n, J, device):
    from hugs_amd.lbs import lbs_skin
    A, W, v, R, r = _body(n, J, seed=n + J)
    gv, gT, gR = (r.standard_normal(s).astype(np.float32) for s in ((n, 3), (n, 4, 4), (n, 3, 3)))
    d = lambda a, grad=False: torch.from_numpy(a).to(device).requires_grad_(grad)
    tA, tW, tv, tR = d(A, True), d(W, True), d(v, True), d(R, True)
    verts, T, rot = lbs_skin(tA, tW, tv, tR)
    rv, rT, rrot = lo.skin(A, W, v, R)
    np.testing.assert_allclose(verts.detach().cpu().numpy(), rv, **TOL)
    np.testing.assert_allclose(T.detach().cpu().numpy(), rT, **TOL)
    np.testing.assert_allclose(rot.detach().cpu().numpy(), rrot, **TOL)
    ((verts * d(gv)).sum() + (T * d(gT)).sum() + (rot * d(gR)).sum()).backward()
    ref = lo.skin_backward(A, W, v, R, gv, gT, gR)
    for x, k in ((tA, "A"), (tW, "weights"), (tv, "v"), (tR, "rotmat")):
        err = np.abs(x.grad.cpu().numpy().astype(np.float64) - ref[k]).max()
        assert err <= 1e-4 * np.abs(ref[k]).max(), (k, err)       # dL/dA sums n products in fp32 (MFMA, fixed order)
    # deterministic: no float atomics anywhere in the backward
    tA2, tW2, tv2, tR2 = d(A, True), d(W, True), d(v, True), d(R, True)
    v2, T2, r2 = lbs_skin(tA2, tW2, tv2, tR2)
    ((v2 * d(gv)).sum() + (T2 * d(gT)).sum() + (r2 * d(gR)).sum()).backward()
    assert torch.equal(tA.grad, tA2.grad) and torch.equal(tW.grad, tW2.grad)
    # gradients only for what was asked: verts alone (dL/dT, dL/drot absent)
    tA3, tW3, tv3 = d(A, True), d(W, True), d(v, True)
    v3, _, _ = lbs_skin(tA3, tW3, tv3, None)
    (v3 * d(gv)).sum().backward()
    ref3 = lo.skin_backward(A, W, v, None, gv)
    assert np.abs(tA3.grad.cpu().numpy() - ref3["A"]).max() <= 1e-4 * np.abs(ref3["A"]).max()


@pytest.mark.gpu
def test_hip_lbs_errors(device):
    from hugs_amd.lbs import lbs_skin
    with pytest.raises(RuntimeError):
        lbs_skin(torch.zeros(24, 4, 4), torch.zeros(5, 24), torch.zeros(5, 3))                 # CPU tensors: no fallback
    with pytest.raises((RuntimeError, ValueError)):
        lbs_skin(torch.zeros(40, 4, 4, device=device), torch.zeros(5, 40, device=device), torch.zeros(5, 3, device=device))   # J > 32
