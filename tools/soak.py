#!/usr/bin/env python3
"""Stability soak on one MI355X: random scene sizes / image shapes / SH degrees, frames alternating between three HIP
streams, forced capacity-guess misses (too-small and absent hints), forward-only and forward+backward, for 40 s.
Prints "soak ok: <frames> ..." or raises.   python tools/soak.py"""
import sys, math, time, random, numpy as np, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-hugs_amd"))
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from hugs_amd import synthetic as syn
dev = torch.device("cuda:0")
random.seed(0)
shapes = [(64, 96), (128, 128), (270, 480), (1080, 1920), (100, 37)]
scenes = {}
def scene(P, H, W):
    k = (P, H, W)
    if k not in scenes:
        cam = syn.pinhole_camera(H, W)
        g = syn.scene_gaussians(P, cam, seed=P % 97, sigma_px=random.choice([2.0, 5.0, 12.0]), ref_P=max(P, 1000), cluster=random.choice([0.0, 0.6]))
        d = lambda a, grad=False: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev).requires_grad_(grad)
        t = {kk: d(g[kk], True) for kk in ("means3D", "opacities", "shs", "scales", "rotations")}
        st = GaussianRasterizationSettings(H, W, math.tan(cam["fovx"]/2), math.tan(cam["fovy"]/2), torch.ones(3, device=dev), 1.0,
             d(cam["world_view_transform"]), d(cam["full_proj_transform"]), random.choice([0, 1, 3]), d(cam["camera_center"]), False, False)
        scenes[k] = (t, st, torch.zeros(P, 3, device=dev, requires_grad=True), torch.randn(3, H, W, device=dev) * 1e-3)
    return scenes[k]
streams = [torch.cuda.current_stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
t0 = time.time(); n = 0
orig_hint = dgr._capacity_hint
while time.time() - t0 < 40:
    H, W = random.choice(shapes); P = random.choice([1, 50, 3000, 40000, 200000 if H >= 270 else 5000])
    t, st, m2d, dL = scene(P, H, W)
    r = random.random()
    dgr._capacity_hint = (lambda k: (64, 1)) if r < 0.1 else ((lambda k: (0, 0)) if r < 0.2 else orig_hint)
    s = random.choice(streams)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        color, radii = GaussianRasterizer(st)(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
        if random.random() < 0.8:
            color.backward(dL)
            for x in list(t.values()) + [m2d]:
                x.grad = None
    n += 1
    if n % 200 == 0:
        torch.cuda.synchronize()
        assert torch.isfinite(color).all()
torch.cuda.synchronize()
print(f"soak ok: {n} frames in {time.time() - t0:.1f} s over {len(scenes)} scenes")
