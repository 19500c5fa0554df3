#!/usr/bin/env python3
"""Densification churn soak: the number of Gaussians changes on EVERY frame (a trainer that clones / splits / prunes:
/root/reference/hugs/models/scene.py:441-458), so every frame is a (P, H, W) shape the hint tables have never seen --
more than 256 distinct shapes per run, which is where the tables clear themselves -- on three streams, and every frame is
rendered by BOTH bindings (C++ autograd node, Python ctypes): the two images must be bit-equal and the gradients within
1e-4.  A slice of the frames goes through deferred render_batch loops.  SOAK_SECONDS (default 60).
    python tools/soak_churn.py   ->   "churn soak ok: ..." or raises."""
import math
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-hugs_amd"))
import diff_gaussian_rasterization as dgr                                                     # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer     # noqa: E402
from hugs_amd import synthetic as syn                                                         # noqa: E402
from hugs_amd.renderer import render, render_batch                                            # noqa: E402

dev = torch.device("cuda:0")
random.seed(int(os.environ.get("SOAK_SEED", "0")))
cpp = dgr._load_cpp()
assert cpp is not None, "the C++ binding is needed (python __graft_entry__.py)"
shapes = [(64, 96), (128, 128), (270, 480), (512, 512), (1080, 1920)]
pools = {}
for H, W in shapes:
    cam = syn.pinhole_camera(H, W)
    Pmax = 60_000 if H >= 512 else 8_000
    g = syn.scene_gaussians(Pmax, cam, seed=H, sigma_px=5.0, ref_P=Pmax)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev)
    pools[(H, W)] = ({k: d(g[k]) for k in ("means3D", "opacities", "shs", "scales", "rotations")},
                     {k: (d(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}, cam, Pmax)
streams = [torch.cuda.current_stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
budget = float(os.environ.get("SOAK_SECONDS", "60"))
t0 = time.time()
seen, n, n_batch = set(), 0, 0
while time.time() - t0 < budget:
    H, W = random.choice(shapes)
    pool, data, cam, Pmax = pools[(H, W)]
    P = random.randint(1, Pmax)                       # a new size practically every time
    seen.add((P, H, W))
    deg = random.choice([0, 1, 3])
    st = GaussianRasterizationSettings(H, W, math.tan(cam["fovx"] / 2), math.tan(cam["fovy"] / 2), torch.ones(3, device=dev), 1.0,
                                       data["world_view_transform"], data["full_proj_transform"], deg, data["camera_center"], False, False)
    s = random.choice(streams)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        results = []
        with_grad = random.random() < 0.6
        for use_cpp in (True, False):
            dgr._cpp, dgr._CPP_WANTED = (cpp, True) if use_cpp else (None, False)
            t = {k: v[:P].clone().requires_grad_(with_grad) for k, v in pool.items()}
            m2d = torch.zeros(P, 3, device=dev, requires_grad=with_grad)
            with torch.set_grad_enabled(with_grad):
                color, radii = GaussianRasterizer(st)(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], shs=t["shs"],
                                                      scales=t["scales"], rotations=t["rotations"])
                if with_grad:
                    color.backward(torch.full_like(color, 1e-3))
            results.append((color.detach(), radii, [t[k].grad for k in ("means3D", "opacities", "scales")] if with_grad else None))
        (c0, r0, g0), (c1, r1, g1) = results
        assert torch.isfinite(c0).all() and torch.equal(c0, c1) and torch.equal(r0, r1), f"bindings disagree on {(P, H, W)}"
        if g0 is not None:
            for a, b in zip(g0, g1):
                assert float((a - b).norm() / b.norm().clamp_min(1e-30)) <= 1e-4, f"gradients differ on {(P, H, W)}"
        if random.random() < 0.1:                     # the frame loops: deferred frames through the per-stream arenas
            fr = dict(means3D=pool["means3D"][:P], feats=pool["shs"][:P], opacity=pool["opacities"][:P], scales=pool["scales"][:P],
                      rotations=pool["rotations"][:P], data=data, bg_color=st.bg, active_sh_degree=deg)
            want = c0.clamp(0.0, 1.0)                 # (the adapter asks for the reference's clamp, fused)
            for out in render_batch([fr] * 4, num_streams=random.choice([1, 2, 3])):
                assert torch.equal(out["render"], want), f"render_batch differs on {(P, H, W)}"
            n_batch += 4
    n += 1
    if n % 64 == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
lib = dgr._load()
assert len(seen) > 256, f"only {len(seen)} distinct shapes: lengthen SOAK_SECONDS"
print(f"churn soak ok: {n} frames x 2 bindings (+{n_batch} deferred) in {time.time() - t0:.1f} s, {len(seen)} distinct (P, H, W) shapes, "
      f"{lib.hgs_debug_stat(b'tile_counter_entries')} counter arrays kept")
