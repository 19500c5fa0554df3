#!/usr/bin/env python3
"""One HUGS-shaped training step through every row this repository owns, on one MI355X -- the part of
/root/reference/hugs/trainer/gs_trainer.py:226-351 that lies between the networks' outputs and the optimizer:

    human:  rot6d -> matrix -> quaternion (hugs_trimlp.py:418-419), learned-LBS skinning + rotation product (:477-489,517),
            matrix -> quaternion of the deformed rotations (:518), ground-truth LBS weights from the 6 nearest template
            vertices (:480-484, no grad)
    scene:  SceneGS.forward (scene.py:147-160)
    render_human_scene with the separate human render (gs_renderer.py:20-99), random backgrounds
    loss:   0.8 l1 + 0.2 (1 - ssim) on both renders (losses/loss.py:88-107,128-137) + the LBS regulariser (mse to the gt weights)
    backward, densification statistics of both models (gs_trainer.py:406-411,429-435)

110 210 human + 200 000 scene Gaussians, 1920x1080 (BASELINE config C4's sizes).  Timed twice: with the fused HIP rows, and with
the reference's torch statements in their place wherever a statement form exists on this box (the rasterizer itself and the
neighbour search have none here: upstream's CUDA rasterizer and pytorch3d's knn_points are not installable on ROCm; they are the
HIP kernels in both runs).  Prints one JSON line.        python tools/bench_step.py [--steps 30]"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-hugs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from hugs_amd import gaussian_io as gio, losses, synthetic as syn                     # noqa: E402
from hugs_amd.densify import update_densification_stats                               # noqa: E402
from hugs_amd.knn import smpl_lbsweight_top_k                                         # noqa: E402
from hugs_amd.lbs import lbs_skin                                                     # noqa: E402
from hugs_amd.renderer import render_human_scene                                      # noqa: E402
from hugs_amd.rotations import matrix_to_quaternion, rotation_6d_to_matrix            # noqa: E402
from hugs_amd.scene_forward import scene_forward                                      # noqa: E402


def torch_rows():
    """The reference's statements for the same rows (restated, as in the row's own bench tools and tests)."""
    from bench_lbs import torch_statements as lbs_statements
    from bench_rotations import torch_6d, torch_m2q
    from test_losses import _torch_statements as loss_statements

    def densify(max_radii2D, accum, denom, vsp, vis, radii):          # scene.py:460-462 + gs_trainer.py:406-411
        n = vis.shape[0]
        max_radii2D[vis] = torch.max(max_radii2D[vis], radii[vis].float())
        accum[vis] += torch.norm(vsp.grad[:n][vis, :2], dim=-1, keepdim=True)
        denom[vis] += 1

    def photometric(a, b):
        s, l1 = loss_statements(a, b)
        return 0.8 * l1 + 0.2 * (1.0 - s)

    return dict(rot6d=torch_6d, m2q=lambda m: torch_m2q(m), lbs=lbs_statements,
                scene=lambda p, deg: gio.activated({**p, "active_sh_degree": deg}), photometric=photometric, densify=densify)


def fused_rows():
    def photometric(a, b):
        return 0.8 * losses.l1_loss(a, b) + 0.2 * (1.0 - losses.ssim(a, b))

    return dict(rot6d=rotation_6d_to_matrix, m2q=matrix_to_quaternion, lbs=lbs_skin,
                scene=lambda p, deg: scene_forward(p["xyz"], p["scaling"], p["rotation"], p["opacity"], p["features_dc"], p["features_rest"], deg),
                photometric=photometric, densify=update_densification_stats)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=int(os.environ.get("HGS_BENCH_STEPS", 30)))
    ap.add_argument("--only", choices=("fused", "torch"), default=None, help="time one of the two variants only")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    H, W, Ph, Ps, J, Mv = 1080, 1920, 110_210, 200_000, 24, 6890
    r = np.random.default_rng(3)
    cam = syn.pinhole_camera(H, W)
    data = {k: (torch.from_numpy(np.ascontiguousarray(v)).float().to(dev) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    t = lambda x, grad=False: torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(dev).requires_grad_(grad)
    from bench_knn import body_surface
    # human: canonical positions on a body-sized surface, 1.2 m tall, 4 m in front of the camera after posing
    templ = body_surface(Mv, r) * 0.6
    canon = body_surface(Ph, r, noise=0.004) * 0.6
    smpl_w = np.exp(-np.linalg.norm(templ[:, None] - templ[r.choice(Mv, J, replace=False)][None], axis=-1) / 0.05)
    smpl_w = (smpl_w / smpl_w.sum(1, keepdims=True)).astype(np.float32)
    A = np.tile(np.eye(4, dtype=np.float32), (J, 1, 1))
    A[:, :3, :3] += 0.05 * r.standard_normal((J, 3, 3)).astype(np.float32)
    A[:, 2, 3] = 4.0
    human = {"xyz": t(canon, True), "rot6d": t(r.standard_normal((Ph, 6)), True), "lbs_logits": t(r.standard_normal((Ph, J)), True),
             "scales": t(0.012 * np.exp(0.3 * r.standard_normal((Ph, 3))), True), "shs": t(0.3 * r.standard_normal((Ph, 16, 3)), True),
             "opacity": t(r.uniform(0.05, 1.0, (Ph, 1)), True)}
    A_t, templ_t, smpl_w_t = t(A, True), t(templ)[None], t(smpl_w)
    g = syn.scene_gaussians(Ps, cam, seed=8, sigma_px=4.0)
    scene = {"xyz": t(g["means3D"], True), "scaling": t(np.log(g["scales"]), True), "rotation": t(g["rotations"], True),
             "opacity": t(np.log(g["opacities"] / np.maximum(1 - g["opacities"], 1e-4)), True),
             "features_dc": t(g["shs"][:, :1], True), "features_rest": t(g["shs"][:, 1:], True)}
    gt1, gt2 = torch.rand(3, H, W, device=dev), torch.rand(3, H, W, device=dev)
    stats = {k: (torch.zeros(n, device=dev), torch.zeros(n, 1, device=dev), torch.zeros(n, 1, device=dev)) for k, n in (("h", Ph), ("s", Ps))}
    leaves = list(human.values()) + list(scene.values()) + [A_t]

    def step(rows):
        rotmat = rows["rot6d"](human["rot6d"])
        rotq_canon = rows["m2q"](rotmat)                                                   # hugs_trimlp.py:419 (an output of the model)
        lbs_w = torch.softmax(human["lbs_logits"] / 0.1, dim=-1)                           # :432 (the decoder's head: torch in both runs)
        xyz, lbs_T, rot_def = rows["lbs"](A_t, lbs_w, human["xyz"], rotmat.reshape(-1, 3, 3) if rotmat.ndim == 2 else rotmat)
        rotq = rows["m2q"](rot_def)
        with torch.no_grad():
            _, gt_w = smpl_lbsweight_top_k(smpl_w_t, human["xyz"][None], templ_t)           # :480-484
        h_out = {"xyz": xyz, "scales": human["scales"], "rotq": rotq, "shs": human["shs"], "opacity": human["opacity"], "active_sh_degree": 0}
        s_out = rows["scene"](scene, 3)
        pkg = render_human_scene(data, h_out, s_out, bg_color=torch.rand(3, device=dev), human_bg_color=torch.rand(3, device=dev),
                                 render_mode="human_scene", render_human_separate=True)
        loss = rows["photometric"](pkg["render"], gt1) + rows["photometric"](pkg["human_img"], gt2) + \
            1000.0 * torch.nn.functional.mse_loss(lbs_w, gt_w[0]) + 1e-3 * rotq_canon.square().mean()
        loss.backward()
        vsp = pkg["viewspace_points"]
        rows["densify"](*stats["h"], vsp, pkg["human_visibility_filter"], pkg["human_radii"])
        rows["densify"](*stats["s"], pkg.get("scene_viewspace_points", vsp), pkg["scene_visibility_filter"], pkg["scene_radii"])
        for x in leaves:
            x.grad = None
        return loss.detach()

    out = {"workload": f"HUGS-shaped step between the networks and the optimizer: {Ph} human + {Ps} scene Gaussians, {W}x{H}, two renders"}
    variants = (("fused_rows", fused_rows, a.steps, "fused"), ("torch_statements_where_they_exist", torch_rows, max(a.steps // 5, 3), "torch"))
    for name, make_rows, n, key in variants:
        if a.only not in (None, key):
            continue
        rows = make_rows()
        for _ in range(3):
            step(rows)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            last = step(rows)
        torch.cuda.synchronize()
        out[name + "_ms_per_step"] = round((time.perf_counter() - t0) / n * 1e3, 3)
        out[name + "_loss"] = round(last.item(), 5)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
