"""SURVEY.md 8f row f-1 -- fused densification statistics.
CPU: the numpy oracle against vectors produced by the reference's own statements (gs_trainer.py:406-411,
scene.py:460-462, executed from /root/reference by tests/golden/make_golden.py).  GPU: the HIP kernel against both."""
import os

import numpy as np
import pytest
import torch

from oracle import densify_oracle as do

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_substeps.npz"))


def test_oracle_matches_reference_statements():
    m, a, d = do.update(G["dens_in_max_radii2D"], G["dens_in_accum"], G["dens_in_denom"], G["dens_grad"], G["dens_vis"],
                        G["dens_radii"])
    assert np.array_equal(m, G["dens_out_max_radii2D"])
    np.testing.assert_allclose(a, G["dens_out_accum"], rtol=1e-6, atol=0)  # fp32: torch.norm and sqrt(x*x+y*y) may differ in the last ulp
    assert np.array_equal(d, G["dens_out_denom"])
    # the gradient tensor is longer than the model (joint mode): only its first n rows are paired with the model
    assert G["dens_grad"].shape[0] > G["dens_vis"].shape[0]


@pytest.mark.gpu
def test_hip_kernel_matches_reference_and_oracle(device):
    from hugs_amd.densify import update_densification_stats
    t = lambda k: torch.from_numpy(G[k].copy()).to(device)
    m, a, d = t("dens_in_max_radii2D"), t("dens_in_accum"), t("dens_in_denom")
    vpt = torch.zeros(G["dens_grad"].shape, device=device, requires_grad=True)
    vpt.grad = t("dens_grad")
    update_densification_stats(m, a, d, vpt, t("dens_vis"), t("dens_radii"))
    assert np.array_equal(m.cpu().numpy(), G["dens_out_max_radii2D"])
    np.testing.assert_allclose(a.cpu().numpy(), G["dens_out_accum"], rtol=1e-6, atol=0)  # fp32: torch.norm and sqrt(x*x+y*y) may differ in the last ulp
    assert np.array_equal(d.cpu().numpy(), G["dens_out_denom"])

    # a larger random case against the oracle, applied twice (the statistics accumulate over steps)
    rng = np.random.default_rng(3)
    n = 100_003
    grad = rng.standard_normal((n + 17, 3)).astype(np.float32)
    radii = rng.integers(0, 80, n).astype(np.int32)
    vis = radii > 5
    m0, a0, d0 = np.zeros(n, np.float32), np.zeros((n, 1), np.float32), np.zeros((n, 1), np.float32)
    rm, ra, rd = do.update(*do.update(m0, a0, d0, grad, vis, radii), grad, vis, radii // 2)
    gm, ga, gd = (torch.from_numpy(x.copy()).to(device) for x in (m0, a0, d0))
    vpt = torch.zeros(n + 17, 3, device=device, requires_grad=True)
    vpt.grad = torch.from_numpy(grad).to(device)
    update_densification_stats(gm, ga, gd, vpt, torch.from_numpy(vis).to(device), torch.from_numpy(radii).to(device))
    update_densification_stats(gm, ga, gd, vpt, torch.from_numpy(vis).to(device), torch.from_numpy(radii // 2).to(device))
    assert np.array_equal(gm.cpu().numpy(), rm) and np.array_equal(gd.cpu().numpy(), rd)
    np.testing.assert_allclose(ga.cpu().numpy(), ra, rtol=1e-6)


@pytest.mark.gpu
def test_densification_stats_after_a_real_render(device):
    """End of a training step as the trainer runs it: render -> backward -> statistics from radii and
    viewspace_points.grad (the NDC-scaled screen-space gradient of SURVEY.md A.6 quirk 6)."""
    from hugs_amd.densify import update_densification_stats
    from hugs_amd.renderer import render
    from scenes import make_scene
    from test_gpu_parity import to_dev
    sc = make_scene(P=500, H=64, W=96, seed=60, D=1)
    cam = {k: (to_dev(v, device) if isinstance(v, np.ndarray) else v) for k, v in sc["cam"].items()}
    x = {k: to_dev(sc[k], device, True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    pkg = render(x["means3D"], x["shs"], x["opacities"], x["scales"], x["rotations"], cam, active_sh_degree=1)
    pkg["render"].sum().backward()
    n = 500
    m, a, d = torch.zeros(n, device=device), torch.zeros(n, 1, device=device), torch.zeros(n, 1, device=device)
    update_densification_stats(m, a, d, pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"])
    vis = pkg["visibility_filter"]
    ref_a = torch.zeros(n, 1, device=device)
    ref_a[vis] += torch.norm(pkg["viewspace_points"].grad[:n][vis, :2], dim=-1, keepdim=True)
    assert torch.allclose(a, ref_a, rtol=1e-6, atol=0)
    assert torch.equal(d.squeeze(1) > 0, vis) and torch.equal(m, torch.where(vis, pkg["radii"].float(), torch.zeros_like(m)))
