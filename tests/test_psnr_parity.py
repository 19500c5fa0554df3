"""north_star's acceptance criterion as far as it goes without the captures: "PSNR within 0.1 dB of the reference" after training.
NeuMan/lab is not here (no dataset, no network: DESIGN.md section 7), so the criterion is run on a synthetic fit of the same SHAPE
as a HUGS scene fit (gs_trainer.py:218-391 in miniature), at BASELINE configs[0]'s size -- 10 000 Gaussians, 256x256 --:

  * the loss of hugs/losses/loss.py:88-107 with the release weights (0.8 l1 + 0.2 (1 - ssim)), Adam with the reference's per-group
    learning rates, 300 steps;
  * densification as the trainer does it (scene.py:441-458, hugs_scene.yaml:111-115 in miniature): gradient statistics from
    `viewspace_points.grad` / radii / visibility on every step, ONE clone / split / prune round and ONE opacity reset on the way;
  * the target is the render of a DIFFERENT scene (six times as many Gaussians of 0.4 of the size, another seed): the model cannot
    reproduce it and the fit ends near 30 dB -- the range NeuMan fits end in -- not at the 47 dB of a fit towards its own render.
    (Calibrated in round 5, DESIGN_HISTORY.md: sigma 0.8 / 1.2 / 1.6 px -> 25.6 / 29.7 / 33.5 dB; four HIP runs spread 0.02-0.13 dB.)

The SAME fit -- same initial parameters, same optimiser, same schedule, same split noise -- is run once with the CPU oracle
(oracle/hgs_oracle.c, the restatement of the published algorithm, + the reference's loss statements in torch) as the rasterizer and
THREE times through the product path (HIP rasterizer under the drop-in API + the fused loss + the fused densification statistics).
Runs of the HIP path differ among themselves: their float atomics land in another order, Adam amplifies it, and the densification
round turns it into another set of Gaussians (/root/reference/README.md:122 warns of the same for the CUDA rasterizer).  So the
statistic the criterion needs is: |mean over the HIP runs - the oracle's run| <= 0.1 dB at the end, with the HIP runs' own spread
reported beside it.  The oracle is the checker here, never the product path."""
import numpy as np
import pytest
import torch

from hugs_amd import metrics, synthetic as syn
from oracle import hgs_oracle as orc

pytestmark = pytest.mark.gpu

H = W = 256
P_MODEL, P_TARGET, DEGREE = 10_000, 60_000, 3
STEPS, DENSIFY_AT, RESET_AT, CHECK_EVERY = 300, 150, 60, 50
HIP_RUNS = 3
LR = {"xyz": 1.6e-4 * 5.0, "features_dc": 2.5e-3, "features_rest": 2.5e-3 / 20.0, "opacity": 5e-2, "scaling": 5e-3, "rotation": 1e-3}   # hugs_scene.yaml:104-110 (position: x the scene extent)


class _OracleRasterizer(torch.autograd.Function):
    """the CPU oracle behind the renderer's contract: (colour [3,H,W], radii) from (means3D, means2D sink, shs, opacities, scales, rotations)"""

    @staticmethod
    def forward(ctx, means3D, means2D, shs, opacities, scales, rotations, cam, bg, degree):
        n = lambda x: x.detach().cpu().numpy()
        inp = orc.Inputs(n(means3D), n(opacities).reshape(-1), cam["world_view_transform"].reshape(-1), cam["full_proj_transform"].reshape(-1),
                         cam["camera_center"], np.tan(cam["fovx"] * 0.5), np.tan(cam["fovy"] * 0.5), cam["image_height"],
                         cam["image_width"], n(bg), shs=n(shs), scales=n(scales), rotations=n(rotations), sh_degree=degree)
        fwd = orc.forward(inp)
        ctx.inp, ctx.fwd = inp, fwd
        radii = torch.from_numpy(fwd["radii"].astype(np.int32))
        ctx.mark_non_differentiable(radii)
        return torch.from_numpy(fwd["color"].copy()), radii

    @staticmethod
    def backward(ctx, g, _):
        gr = orc.backward(ctx.inp, ctx.fwd, g.contiguous().numpy())
        t = torch.from_numpy
        return t(gr["means3D"]), t(gr["means2D"]), t(gr["shs"]), t(gr["opacities"]), t(gr["scales"]), t(gr["rotations"]), None, None, None


def _render_hip(act, data, bg):
    from hugs_amd.renderer.gs_renderer import render
    return render(means3D=act["xyz"], feats=act["shs"], opacity=act["opacity"], scales=act["scales"], rotations=act["rotq"], data=data,
                  bg_color=bg, active_sh_degree=DEGREE)


def _render_oracle(act, cam, bg):
    vs = torch.zeros_like(act["xyz"], requires_grad=True)
    color, radii = _OracleRasterizer.apply(act["xyz"], vs, act["shs"], act["opacity"], act["scales"], act["rotq"], cam, bg, DEGREE)
    return {"render": color.clamp(0.0, 1.0), "viewspace_points": vs, "radii": radii, "visibility_filter": radii > 0}   # gs_renderer.py:153-160


def _loss_hip(pred, target):
    from hugs_amd.losses import l1_loss, ssim
    return 0.8 * l1_loss(pred, target) + 0.2 * (1.0 - ssim(pred, target))


def _loss_statements(pred, target):
    """hugs/losses/utils.py:54-58,65-108 restated in torch (tests/test_losses.py holds them against the fused kernels)"""
    from test_losses import _torch_statements
    s, l1 = _torch_statements(pred, target)
    return 0.8 * l1 + 0.2 * (1.0 - s)


def _stats_statements(gs, pkg):
    """gs_trainer.py:406-411 / scene.py:460-462 as the reference states them (the oracle fit's side of hugs_amd.densify)"""
    vis, g = pkg["visibility_filter"], pkg["viewspace_points"].grad
    gs.max_radii2D[vis] = torch.max(gs.max_radii2D[vis], pkg["radii"][vis].float())
    gs.grad_accum[vis] += torch.norm(g[vis, :2], dim=-1, keepdim=True)
    gs.denom[vis] += 1


def _fit(kind, init, target, cam, device, split_seed):
    """-> (PSNR every CHECK_EVERY steps, number of Gaussians at the end)"""
    import test_hugs_loop as loop
    loop_lr, loop.LR = loop.LR, LR    # (GaussianSet rebuilds its optimiser from the module's table after every growth)
    try:
        return _fit_body(loop, kind, init, target, cam, device, split_seed)
    finally:
        loop.LR = loop_lr


def _fit_body(loop, kind, init, target, cam, device, split_seed):
    from hugs_amd.densify import update_densification_stats
    gs = loop.GaussianSet({k: torch.from_numpy(v.copy()).to(device) for k, v in init.items()}, DEGREE)
    gs.noise_gen = torch.Generator(device="cpu").manual_seed(split_seed)    # the same split noise in every fit
    target = target.to(device)
    bg = torch.ones(3, device=device)
    data = {k: (torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(device) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    draw = (lambda: _render_hip(gs.activated(), data, bg)) if kind == "hip" else (lambda: _render_oracle(gs.activated(), cam, bg))
    loss_fn = _loss_hip if kind == "hip" else _loss_statements
    curve = []
    for step in range(STEPS + 1):
        if step % CHECK_EVERY == 0:
            with torch.no_grad():
                curve.append(float(metrics.psnr(draw()["render"][None], target[None]).mean()))
        if step == STEPS:
            break
        gs.opt.zero_grad(set_to_none=True)
        pkg = draw()
        loss_fn(pkg["render"], target).backward()
        with torch.no_grad():
            if kind == "hip":
                update_densification_stats(gs.max_radii2D, gs.grad_accum, gs.denom, pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"])
            else:
                _stats_statements(gs, pkg)
        gs.opt.step()
        if step + 1 == DENSIFY_AT:
            gs.densify_and_prune(0.0002, 0.005, extent=5.0, max_screen_size=None)
        if step + 1 == RESET_AT:     # scene.py:reset_opacity: opacities capped at 0.01, Adam's moments of the group restart
            with torch.no_grad():
                capped = torch.minimum(gs.p["opacity"], torch.full_like(gs.p["opacity"], float(np.log(0.01 / 0.99))))
            gs._rebuild(lambda k, t, is_moment: (torch.zeros_like(t) if is_moment else capped) if k == "opacity" else t)
    assert all(torch.isfinite(v).all() for v in gs.p.values())
    return curve, int(gs.p["xyz"].shape[0])


def test_fits_through_the_hip_path_end_at_the_psnr_of_the_same_fit_through_the_oracle(device):
    cam = syn.pinhole_camera(H, W)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    tgt = syn.scene_gaussians(P_TARGET, cam, seed=77, sigma_px=1.2, ref_P=P_TARGET)   # six times the model's Gaussians at 0.4 of their size
    g = syn.scene_gaussians(P_MODEL, cam, seed=78, sigma_px=3.0, ref_P=P_MODEL)
    r = np.random.default_rng(9)
    op = np.clip(g["opacities"], 0.02, 0.98)
    init = {"xyz": f32(g["means3D"]), "features_dc": f32(0.3 * r.standard_normal((P_MODEL, 1, 3))), "features_rest": np.zeros((P_MODEL, 15, 3), np.float32),
            "opacity": f32(np.log(op / (1 - op))).reshape(-1, 1), "scaling": f32(np.log(g["scales"])), "rotation": f32(g["rotations"])}
    orc.set_threads(orc.usable_cpus())
    inp = orc.Inputs(f32(tgt["means3D"]), f32(tgt["opacities"]).reshape(-1), cam["world_view_transform"].reshape(-1), cam["full_proj_transform"].reshape(-1),
                     cam["camera_center"], np.tan(cam["fovx"] * 0.5), np.tan(cam["fovy"] * 0.5), H, W, np.ones(3, np.float32), shs=f32(tgt["shs"]),
                     scales=f32(tgt["scales"]), rotations=f32(tgt["rotations"]), sh_degree=DEGREE)
    target = torch.from_numpy(np.clip(orc.forward(inp)["color"], 0.0, 1.0).copy())

    hip = [_fit("hip", init, target, cam, device, split_seed=4) for _ in range(HIP_RUNS)]
    oracle_curve, oracle_n = _fit("oracle", init, target, cam, torch.device("cpu"), split_seed=4)

    finals = np.array([c[-1] for c, _ in hip])
    report = (f"oracle {oracle_curve[-1]:.3f} dB ({oracle_n} Gaussians) / HIP runs {np.round(finals, 3).tolist()} dB ({[n for _, n in hip]} Gaussians): "
              f"mean {finals.mean():.3f}, spread {finals.max() - finals.min():.3f} dB; curves: oracle {np.round(oracle_curve, 2).tolist()}, "
              f"HIP {[np.round(c, 2).tolist() for c, _ in hip]}")
    print(report)
    assert oracle_curve[-1] > oracle_curve[0] + 5.0, f"the oracle's own fit did not converge: {report}"
    assert 20.0 <= oracle_curve[-1] <= 40.0, f"the fit is meant to plateau where real captures do: {report}"
    assert min(n for _, n in hip) > P_MODEL * 0.5 and oracle_n != P_MODEL, f"the densification round did nothing: {report}"
    # the criterion: within 0.1 dB at the end; the HIP runs' own spread is the resolution it can be stated with
    assert abs(finals.mean() - oracle_curve[-1]) <= 0.1, report
    # ... and no single run hides behind the mean: up to the densification round the fit is deterministic up to the order of the float
    # atomics, so EVERY run's checkpoints there are within 0.1 dB of the oracle's; afterwards (one Gaussian more or less cloned moves the curve)
    # every run stays within 0.25 dB of it, and the runs within 0.2 dB of each other
    before = [k for k in range(len(oracle_curve)) if k * CHECK_EVERY <= DENSIFY_AT]
    for c, _ in hip:
        for k in before:
            assert abs(c[k] - oracle_curve[k]) <= 0.1, f"checkpoint at step {k * CHECK_EVERY} (before the densification round): {report}"
        for k in range(len(oracle_curve)):
            assert abs(c[k] - oracle_curve[k]) <= 0.25, f"checkpoint at step {k * CHECK_EVERY}: {report}"
    assert finals.max() - finals.min() <= 0.2, f"spread of the HIP runs: {report}"
