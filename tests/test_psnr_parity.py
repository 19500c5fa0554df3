"""north_star's acceptance criterion in miniature: "PSNR within 0.1 dB of the reference" after training.  The NeuMan
captures are not here (no dataset, no network: DESIGN.md section 7), so the criterion is run on a synthetic fit: the SAME
optimisation -- same initial parameters, same Adam, same loss, same number of steps (gs_trainer.py:218-391 in miniature) --
once with the HIP rasterizer under the drop-in API and once with the CPU oracle (oracle/hgs_oracle.c, the restatement of
the published algorithm) as the rasterizer, and the validation PSNR (hugs/utils/image.py:27-29 -> hugs_amd.metrics.psnr)
of the two runs is compared along the way and at the end.  The oracle is the checker here, never the product path."""
import numpy as np
import pytest
import torch

from hugs_amd import metrics, synthetic as syn
from hugs_amd.renderer.gs_renderer import render
from oracle import hgs_oracle as orc

pytestmark = pytest.mark.gpu

# 80 steps take the fit from 20.7 to ~47 dB.  The comparison stops there on purpose: beyond ~48 dB a PSNR difference of 0.1 dB is an
# MSE difference of a few 1e-7 -- less than two runs of the HIP path differ by among themselves (their float atomics land in
# another order, and 100+ Adam steps amplify it: at 120 steps four runs read 50.96 .. 51.46 dB against the oracle's 51.41, while
# up to step 90 all of them stay within 0.04 dB of it).  NeuMan-level PSNRs (25-35 dB) are well inside the compared range.
STEPS, CHECK_EVERY = 80, 20


class _OracleRasterizer(torch.autograd.Function):
    """the CPU oracle behind the same autograd contract: colour [3,H,W] from (means3D, shs, opacities, scales, rotations)"""

    @staticmethod
    def forward(ctx, means3D, shs, opacities, scales, rotations, cam, bg, degree):
        n = lambda x: x.detach().cpu().numpy()
        inp = orc.Inputs(n(means3D), n(opacities).reshape(-1), cam["world_view_transform"].reshape(-1), cam["full_proj_transform"].reshape(-1),
                         cam["camera_center"], np.tan(cam["fovx"] * 0.5), np.tan(cam["fovy"] * 0.5), cam["image_height"],
                         cam["image_width"], n(bg), shs=n(shs), scales=n(scales), rotations=n(rotations), sh_degree=degree)
        fwd = orc.forward(inp)
        ctx.inp, ctx.fwd = inp, fwd
        return torch.from_numpy(fwd["color"].copy())

    @staticmethod
    def backward(ctx, g):
        gr = orc.backward(ctx.inp, ctx.fwd, g.contiguous().numpy())
        t = torch.from_numpy
        return t(gr["means3D"]), t(gr["shs"]), t(gr["opacities"]), t(gr["scales"]), t(gr["rotations"]), None, None, None


def _fit(draw, init, target, device):
    """the loop of tests/test_training_loop.py; returns the PSNR every CHECK_EVERY steps and the final image"""
    params = {k: torch.from_numpy(v.copy()).to(device).requires_grad_(True) for k, v in init.items()}
    opt = torch.optim.Adam([{"params": [params["xyz"]], "lr": 2e-3}, {"params": [params["dc"]], "lr": 2e-2},
                            {"params": [params["rest"]], "lr": 1e-3}, {"params": [params["opacity"]], "lr": 3e-2},
                            {"params": [params["scaling"]], "lr": 5e-3}, {"params": [params["rotation"]], "lr": 1e-3}])
    target = target.to(device)
    curve = []
    for step in range(STEPS + 1):
        if step % CHECK_EVERY == 0:
            with torch.no_grad():
                curve.append(float(metrics.psnr(draw(params), target).mean()))
        if step == STEPS:
            break
        opt.zero_grad(set_to_none=True)
        loss = (draw(params) - target).abs().mean()
        loss.backward()
        opt.step()
    with torch.no_grad():
        final = draw(params).cpu()
    assert all(torch.isfinite(v).all() for v in params.values())
    return curve, final


def test_a_fit_through_the_hip_rasterizer_reaches_the_psnr_of_the_same_fit_through_the_oracle(device):
    H, W, P, degree = 96, 128, 600, 3
    cam = syn.pinhole_camera(H, W)
    g = syn.scene_gaussians(P, cam, seed=31, sigma_px=5.0, ref_P=P)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    truth = {"xyz": f32(g["means3D"]), "dc": f32(g["shs"][:, :1]), "rest": f32(g["shs"][:, 1:]),
             "opacity": f32(np.log(np.clip(g["opacities"], 1e-3, 1 - 1e-3) / (1 - np.clip(g["opacities"], 1e-3, 1 - 1e-3)))),
             "scaling": f32(np.log(g["scales"])), "rotation": f32(g["rotations"])}
    r = np.random.default_rng(3)
    noisy = lambda x, s: f32(x + s * r.standard_normal(x.shape))
    init = {"xyz": noisy(truth["xyz"], 0.02), "dc": noisy(truth["dc"], 0.5), "rest": np.zeros_like(truth["rest"]),
            "opacity": noisy(truth["opacity"], 0.5), "scaling": noisy(truth["scaling"], 0.2), "rotation": noisy(truth["rotation"], 0.1)}
    bg_np = np.ones(3, np.float32)

    def activated(p):
        return (p["xyz"], torch.cat((p["dc"], p["rest"]), 1), torch.sigmoid(p["opacity"]), torch.exp(p["scaling"]),
                torch.nn.functional.normalize(p["rotation"]))

    data = {k: (torch.from_numpy(f32(v)).to(device) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    bg_gpu, bg_cpu = torch.from_numpy(bg_np).to(device), torch.from_numpy(bg_np)

    def draw_hip(p):
        xyz, feats, op, sc, rot = activated(p)
        return render(means3D=xyz, feats=feats, opacity=op, scales=sc, rotations=rot, data=data, bg_color=bg_gpu,
                      active_sh_degree=degree)["render"]

    def draw_oracle(p):
        xyz, feats, op, sc, rot = activated(p)
        return _OracleRasterizer.apply(xyz, feats, op, sc, rot, cam, bg_cpu, degree).clamp(0.0, 1.0)   # gs_renderer.py:153

    orc.set_threads(8)      # (a 96x128 frame: more threads than tiles only cost their start-up)
    with torch.no_grad():   # one target for both runs: the oracle's render of the truth
        target = draw_oracle({k: torch.from_numpy(v) for k, v in truth.items()})
    curve_hip, img_hip = _fit(draw_hip, init, target, device)
    curve_orc, img_orc = _fit(draw_oracle, init, target, torch.device("cpu"))

    assert curve_orc[-1] > curve_orc[0] + 6.0, f"the oracle's own fit did not converge: {curve_orc}"
    # the criterion: within 0.1 dB, at every validation point along the way and at the end
    for a, b in zip(curve_hip, curve_orc):
        assert abs(a - b) <= 0.1, f"PSNR curves part: HIP {curve_hip} / oracle {curve_orc}"
    # ... and the two fits arrive at the same image (rounding differences amplified by 80 Adam steps stay small)
    assert float(metrics.psnr(img_hip, img_orc).mean()) > 40.0
