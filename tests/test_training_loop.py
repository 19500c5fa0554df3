"""End to end: the rasterizer inside an optimisation loop, the way the HUGS trainer uses it (render -> image loss ->
backward -> Adam on positions, log-scales, quaternions, opacity logits and SH; gs_trainer.py:218-391 in miniature).
Gradients that pass the oracle checks but pointed the wrong way, or were mis-scaled between tensors, would not fit."""
import math

import numpy as np
import pytest
import torch

from hugs_amd import metrics, synthetic as syn
from hugs_amd.renderer.gs_renderer import render

pytestmark = pytest.mark.gpu


def test_fitting_a_target_image_raises_psnr(device):
    H, W, P = 128, 160, 1500
    cam = syn.pinhole_camera(H, W)
    g = syn.scene_gaussians(P, cam, seed=21, sigma_px=6.0, ref_P=P)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(device)
    data = {k: (t(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    bg = torch.ones(3, device=device)

    def draw(p):
        return render(means3D=p["xyz"], feats=torch.cat((p["dc"], p["rest"]), 1), opacity=torch.sigmoid(p["opacity"]),
                      scales=torch.exp(p["scaling"]), rotations=torch.nn.functional.normalize(p["rotation"]), data=data,
                      bg_color=bg, active_sh_degree=3)

    truth = {"xyz": t(g["means3D"]), "dc": t(g["shs"][:, :1]), "rest": t(g["shs"][:, 1:]),
             "opacity": torch.logit(t(g["opacities"]).clamp(1e-3, 1 - 1e-3)), "scaling": torch.log(t(g["scales"])),
             "rotation": t(g["rotations"])}
    with torch.no_grad():
        target = draw(truth)["render"]
    gen = torch.Generator(device="cpu").manual_seed(3)
    noise = lambda x, s: x + s * torch.randn(x.shape, generator=gen).to(device)
    params = {"xyz": noise(truth["xyz"], 0.02), "dc": noise(truth["dc"], 0.5), "rest": torch.zeros_like(truth["rest"]),
              "opacity": noise(truth["opacity"], 0.5), "scaling": noise(truth["scaling"], 0.2), "rotation": noise(truth["rotation"], 0.1)}
    params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    opt = torch.optim.Adam([{"params": [params["xyz"]], "lr": 2e-3}, {"params": [params["dc"]], "lr": 2e-2},
                            {"params": [params["rest"]], "lr": 1e-3}, {"params": [params["opacity"]], "lr": 3e-2},
                            {"params": [params["scaling"]], "lr": 5e-3}, {"params": [params["rotation"]], "lr": 1e-3}])
    psnr0 = float(metrics.psnr(draw(params)["render"].detach(), target).mean())
    for _ in range(150):
        opt.zero_grad(set_to_none=True)
        pkg = draw(params)
        loss = (pkg["render"] - target).abs().mean()
        loss.backward()
        assert pkg["viewspace_points"].grad is not None      # the densification signal is there every step
        opt.step()
    psnr1 = float(metrics.psnr(draw(params)["render"].detach(), target).mean())
    assert all(torch.isfinite(v).all() for v in params.values())
    assert psnr1 > psnr0 + 6.0, f"PSNR {psnr0:.2f} -> {psnr1:.2f} dB"
