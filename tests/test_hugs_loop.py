"""A HUGS-shaped optimisation loop around the rasterizer (the stand-in for BASELINE's PSNR-on-NeuMan criterion, which
needs a dataset this box does not have): everything the trainer does to the rasterizer's inputs and outputs, in miniature.

  * a scene model created from a point cloud with `distCUDA2` scales (/root/reference/hugs/models/scene.py:166-194) and a
    fixed-size human model, both optimised with Adam (scene.py:196-218);
  * every step: a random training camera, a random background and a random human background (gs_trainer.py:254-259), TWO
    renders through `render_human_scene(..., render_human_separate=True)` (gs_renderer.py:56-82) -- the joint one through
    the two-segment form of the C ABI --, the reference's photometric loss on both (0.8 l1 + 0.2 (1 - ssim),
    hugs/losses/loss.py:88-107,128-137, through the fused kernels of row f-5), one backward;
  * the fused densification statistics on the joint render's `viewspace_points.grad` with the scene's filter -- paired, as
    the reference does it, with the FIRST n rows of the gradient (gs_trainer.py:316-327, scene.py:460-462);
  * clone / split / prune every 50 steps with the reference's thresholds (scene.py:400-458: grad 0.0002, percent_dense
    0.01, min opacity 0.005, max_screen_size 20 once past the first opacity-reset interval, gs_trainer.py:406-427), so the
    number of Gaussians -- and with it every scratch size, hint and arena -- changes all the time; `oneupSHdegree` at
    100 / 200 (scene.py:162-164);
  * at the end: PLY save (scene.py:243-260), reload in a FRESH PROCESS, same render bit for bit.

Both bindings.  The densification bookkeeping below is test infrastructure in the test's own words, not product code."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from hugs_amd import gaussian_io as gio, metrics, synthetic as syn
from hugs_amd.densify import update_densification_stats
from hugs_amd.knn import distCUDA2
from hugs_amd.renderer import render_human_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W = 160, 224
LR = {"xyz": 2e-3, "features_dc": 2e-2, "features_rest": 1e-3, "opacity": 4e-2, "scaling": 6e-3, "rotation": 1e-3}


class GaussianSet:
    """Parameters in the reference's storage conventions (logit opacity, log scales, raw quaternions, dc/rest SH) + Adam,
    with the three growth / shrink operations of the trainer: append rows (moments of new rows zero), drop rows."""

    def __init__(self, params, degree, max_degree=3):
        self.p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        self.degree, self.max_degree = degree, max_degree
        self._make_opt({})
        self.reset_stats()

    def _make_opt(self, moments):
        self.opt = torch.optim.Adam([{"params": [self.p[k]], "lr": LR[k], "name": k} for k in LR], lr=0.0, eps=1e-15)
        for k, (m, v, step) in moments.items():
            self.opt.state[self.p[k]] = {"step": step, "exp_avg": m, "exp_avg_sq": v}

    def reset_stats(self):
        n, dev = self.p["xyz"].shape[0], self.p["xyz"].device
        self.grad_accum, self.denom = torch.zeros(n, 1, device=dev), torch.zeros(n, 1, device=dev)
        self.max_radii2D = torch.zeros(n, device=dev)

    noise_gen = None
    fused_forward = False   # the scene model goes through the fused SceneGS.forward of row f-6, the human through the torch statements

    def activated(self):
        if self.fused_forward:
            from hugs_amd.scene_forward import scene_forward
            p = self.p
            return scene_forward(p["xyz"], p["scaling"], p["rotation"], p["opacity"], p["features_dc"], p["features_rest"], self.degree)
        return gio.activated({**self.p, "active_sh_degree": self.degree})

    def _rebuild(self, rows_of):
        """rows_of(name, tensor, is_moment) -> the tensor's new rows; applied to the parameters and to Adam's moments alike"""
        moments = {}
        for k in LR:
            st = self.opt.state.get(self.p[k])
            if st:
                moments[k] = (rows_of(k, st["exp_avg"], True), rows_of(k, st["exp_avg_sq"], True), st["step"])
        self.p = {k: rows_of(k, v.detach(), False).clone().requires_grad_(True) for k, v in self.p.items()}
        self._make_opt(moments)

    def append(self, new):
        self._rebuild(lambda k, t, is_moment: torch.cat((t, torch.zeros_like(new[k]) if is_moment else new[k]), 0))

    def keep(self, mask):
        self._rebuild(lambda k, t, is_moment: t[mask])
        self.grad_accum, self.denom, self.max_radii2D = self.grad_accum[mask], self.denom[mask], self.max_radii2D[mask]

    def densify_and_prune(self, grad_threshold, min_opacity, extent, max_screen_size, percent_dense=0.01):
        """scene.py:441-458 with :400-439 inlined: clone the small high-gradient points, split the large ones in two,
        then prune the transparent / oversized ones."""
        with torch.no_grad():
            g = self.grad_accum / self.denom
            g[g.isnan()] = 0.0
            hot = g.squeeze(1) >= grad_threshold
            small = self.p["scaling"].exp().max(dim=1).values <= percent_dense * extent
            clone = hot & small
            if clone.any():
                self.append({k: v.detach()[clone] for k, v in self.p.items()})
            n_before = hot.shape[0]
            split = torch.zeros(self.p["xyz"].shape[0], dtype=torch.bool, device=hot.device)
            split[:n_before] = hot & ~small          # (clones have no gradient history: never split in the same round)
            if split.any():
                sc = self.p["scaling"].detach()[split].exp().repeat(2, 1)
                q = torch.nn.functional.normalize(self.p["rotation"].detach()[split]).repeat(2, 1)
                w, x, y, z = q.unbind(1)
                R = torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                                 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                                 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)), 1).view(-1, 3, 3)
                # (noise_gen: a CPU generator when two fits on different devices must split alike, tests/test_psnr_parity.py)
                noise = torch.randn(sc.shape, generator=self.noise_gen).to(sc) if self.noise_gen is not None else torch.randn_like(sc)
                offs = torch.bmm(R, (noise * sc).unsqueeze(-1)).squeeze(-1)
                new = {k: v.detach()[split].repeat(2, *([1] * (v.dim() - 1))) for k, v in self.p.items()}
                new["xyz"] = new["xyz"] + offs
                new["scaling"] = torch.log(sc / (0.8 * 2))
                self.append(new)
                keep = torch.cat((~split, torch.ones(2 * int(split.sum()), dtype=torch.bool, device=split.device)))
                self.p_stats_resize()
                self.keep(keep)
            self.p_stats_resize()
            prune = torch.sigmoid(self.p["opacity"].detach()).squeeze(1) < min_opacity
            if max_screen_size:
                prune |= (self.max_radii2D > max_screen_size) | (self.p["scaling"].detach().exp().max(dim=1).values > 0.1 * extent)
            self.keep(~prune)
            self.reset_stats()

    def p_stats_resize(self):
        n = self.p["xyz"].shape[0]
        if self.grad_accum.shape[0] != n:   # (statistics restart after every growth, scene.py:396-398)
            self.reset_stats()


def photometric(pred, target):
    """hugs/losses/loss.py:96,107 with the release weights (l_l1_w 0.8, l_ssim_w 0.2), both terms from the fused f-5 kernels."""
    from hugs_amd.losses import l1_loss, ssim
    return 0.8 * l1_loss(pred, target) + 0.2 * (1.0 - ssim(pred, target))


def build_problem(device, seed=5):
    cam0 = syn.pinhole_camera(H, W)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(device)
    cams = []
    for i in range(6):
        yaw = math.radians(3.0) * (i - 2.5)
        w2c = np.eye(4)
        w2c[0, 0], w2c[0, 2], w2c[2, 0], w2c[2, 2] = math.cos(yaw), math.sin(yaw), -math.sin(yaw), math.cos(yaw)
        c = syn.camera_from_w2c(w2c, cam0["fovx"], cam0["fovy"], H, W)
        cams.append({k: (t(v) if isinstance(v, np.ndarray) else v) for k, v in c.items()})
    g = syn.scene_gaussians(2500, cam0, seed=seed, sigma_px=5.0, ref_P=2500)
    rng = np.random.default_rng(seed)
    Ph = 900
    hq = rng.standard_normal((Ph, 4))
    human_truth = {"xyz": t(rng.standard_normal((Ph, 3)) * np.array([0.25, 0.6, 0.15]) + np.array([0.0, 0.0, 5.0])),
                   "scales": t(0.05 * np.exp(0.3 * rng.standard_normal((Ph, 3)))),
                   "rotq": t(hq / np.linalg.norm(hq, axis=1, keepdims=True)),
                   "shs": t(np.concatenate([rng.standard_normal((Ph, 1, 3)), np.zeros((Ph, 15, 3))], 1)),
                   "opacity": t(rng.uniform(0.3, 1.0, (Ph, 1))), "active_sh_degree": 0}
    scene_truth = {"xyz": t(g["means3D"]), "scales": t(g["scales"]), "rotq": t(g["rotations"]), "shs": t(g["shs"]),
                   "opacity": t(g["opacities"]), "active_sh_degree": 3}
    return cams, human_truth, scene_truth


def initial_models(human_truth, scene_truth, device, seed=11):
    gen = torch.Generator(device="cpu").manual_seed(seed)
    noise = lambda x, s: x + s * torch.randn(x.shape, generator=gen).to(device)
    # the scene starts as a coloured point cloud: half of the true centres, isotropic scales from the mean squared distance to
    # the three nearest neighbours, identity rotations, opacity 0.1 (scene.py:166-194)
    pick = torch.randperm(scene_truth["xyz"].shape[0], generator=gen)[:1200].to(device)
    pts = noise(scene_truth["xyz"][pick], 0.01)
    dist2 = torch.clamp_min(distCUDA2(pts), 1e-7)
    n = pts.shape[0]
    scene = GaussianSet({"xyz": pts, "features_dc": scene_truth["shs"][pick, :1].clone(), "features_rest": torch.zeros(n, 15, 3, device=device),
                         "opacity": torch.logit(torch.full((n, 1), 0.1, device=device)),
                         "scaling": torch.log(torch.sqrt(dist2))[:, None].repeat(1, 3),
                         "rotation": torch.tensor([1.0, 0, 0, 0], device=device).repeat(n, 1)}, degree=0)
    human = GaussianSet({"xyz": noise(human_truth["xyz"], 0.02), "features_dc": noise(human_truth["shs"][:, :1], 0.5),
                         "features_rest": torch.zeros(human_truth["xyz"].shape[0], 15, 3, device=device),
                         "opacity": torch.logit(human_truth["opacity"].clamp(0.02, 0.98)) * 0.5,
                         "scaling": noise(torch.log(human_truth["scales"]), 0.2), "rotation": noise(human_truth["rotq"], 0.1)}, degree=0)
    return human, scene


@pytest.mark.parametrize("binding", ["cpp", "ctypes"])
def test_hugs_shaped_loop(binding, device, monkeypatch, tmp_path):
    import diff_gaussian_rasterization as dgr
    if binding == "ctypes":
        monkeypatch.setattr(dgr, "_cpp", None)
        monkeypatch.setattr(dgr, "_CPP_WANTED", False)
    torch.manual_seed(0)
    cams, human_truth, scene_truth = build_problem(device)
    human, scene = initial_models(human_truth, scene_truth, device)
    scene.fused_forward = True
    val_bg = torch.tensor([0.2, 0.5, 0.8], device=device)

    def validate():
        with torch.no_grad():
            a = render_human_scene(cams[0], human.activated(), scene.activated(), bg_color=val_bg, render_mode="human_scene")["render"]
            b = render_human_scene(cams[0], {**human_truth, "active_sh_degree": human.degree}, scene_truth, bg_color=val_bg,
                                   render_mode="human_scene")["render"]
        return float(metrics.psnr(a, b).mean()), a

    psnr0, _ = validate()
    sizes = []
    for it in range(1, 301):
        cam = cams[int(torch.randint(len(cams), (1,)))]
        bg, hbg = torch.rand(3, device=device), torch.rand(3, device=device)
        with torch.no_grad():   # the "ground truth" frame of this step, on the same backgrounds
            ht = {**human_truth, "active_sh_degree": human.degree}
            tgt = render_human_scene(cam, ht, scene_truth, bg_color=bg, human_bg_color=hbg, render_mode="human_scene",
                                     render_human_separate=True)
        human.opt.zero_grad(set_to_none=True), scene.opt.zero_grad(set_to_none=True)
        pkg = render_human_scene(cam, human.activated(), scene.activated(), bg_color=bg, human_bg_color=hbg,
                                 render_mode="human_scene", render_human_separate=True)
        loss = photometric(pkg["render"], tgt["render"]) + 0.5 * photometric(pkg["human_img"], tgt["human_img"])
        loss.backward()
        assert torch.isfinite(loss) and pkg["viewspace_points"].grad is not None
        n_h, n_s = human.p["xyz"].shape[0], scene.p["xyz"].shape[0]
        assert pkg["radii"].shape[0] == n_h + n_s and pkg["scene_radii"].shape[0] == n_s and pkg["human_radii"].shape[0] == n_h
        # scene densification statistics: the fused f-1 kernel on the joint render's gradient, first n_s rows (as the reference)
        update_densification_stats(scene.max_radii2D, scene.grad_accum, scene.denom, pkg["viewspace_points"],
                                   pkg["scene_visibility_filter"], pkg["scene_radii"])
        human.opt.step(), scene.opt.step()
        if it % 50 == 0 and it < 300:
            scene.densify_and_prune(0.0002, 0.005, extent=3.0, max_screen_size=20 if it > 150 else None)
            sizes.append(scene.p["xyz"].shape[0])
        if it in (100, 200):
            human.degree = min(human.degree + 1, human.max_degree)   # oneupSHdegree: the JOINT render takes the human's degree
    psnr1, final = validate()
    assert all(torch.isfinite(v).all() for m in (human, scene) for v in m.p.values())
    assert len(set(sizes)) >= 4, f"the number of scene Gaussians should keep changing: {sizes}"
    assert psnr1 >= psnr0 + 8.0, f"PSNR {psnr0:.2f} -> {psnr1:.2f} dB, scene sizes {sizes}"

    # PLY round trip through a fresh process: same frame, bit for bit
    for name, m in (("human", human), ("scene", scene)):
        gio.write_gaussian_ply(str(tmp_path / f"{name}.ply"), m.p["xyz"], m.p["features_dc"], m.p["features_rest"], m.p["opacity"],
                               m.p["scaling"], m.p["rotation"])
    np.save(tmp_path / "cam.npy", {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in cams[0].items()}, allow_pickle=True)
    code = f"""
import sys, numpy as np, torch
sys.path.insert(0, {os.path.join(ROOT, 'ml-hugs_amd')!r})
from hugs_amd import gaussian_io as gio
from hugs_amd.renderer import render_human_scene
dev = torch.device({str(device)!r})
cam = np.load({str(tmp_path / 'cam.npy')!r}, allow_pickle=True).item()
cam = {{k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}}
h = gio.activated(gio.read_gaussian_ply({str(tmp_path / 'human.ply')!r}, device=dev)); h["active_sh_degree"] = {human.degree}
from hugs_amd.scene_forward import scene_forward          # the scene's activations: the fused kernel, as in the loop
p = gio.read_gaussian_ply({str(tmp_path / 'scene.ply')!r}, device=dev)
s = scene_forward(p["xyz"], p["scaling"], p["rotation"], p["opacity"], p["features_dc"], p["features_rest"], 0)
with torch.no_grad():
    img = render_human_scene(cam, h, s, bg_color=torch.tensor([0.2, 0.5, 0.8], device=dev), render_mode="human_scene")["render"]
np.save({str(tmp_path / 'img.npy')!r}, img.cpu().numpy())
"""
    env = dict(os.environ, HGS_BINDING=binding)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert np.array_equal(np.load(tmp_path / "img.npy"), final.cpu().numpy()), "the reloaded PLYs render a different image"
