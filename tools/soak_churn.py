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
from hugs_amd.renderer import gs_renderer, render, render_batch, render_human_scene           # noqa: E402

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
# (the soak moves the SAME leaves from stream to stream on purpose: torch's AccumulateGrad stream-mismatch warning is about exactly that,
#  and this is the switch it names for an intentional mismatch)
if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
budget = float(os.environ.get("SOAK_SECONDS", "60"))
t0 = time.time()
seen, n, n_batch, n_pair = set(), 0, 0, 0
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
        if P >= 2 and random.random() < 0.15:         # (round 5) both renders of a step as ONE node against the statement path's two
            dgr._cpp, dgr._CPP_WANTED = cpp, True
            cut = random.randint(1, P - 1)
            outs = []
            for frame_call in (True, False):
                gs_renderer._FRAME_CALL = frame_call
                model = lambda lo, hi, d: {"xyz": pool["means3D"][lo:hi].clone().requires_grad_(True), "shs": pool["shs"][lo:hi].clone().requires_grad_(True),
                                           "opacity": pool["opacities"][lo:hi].clone().requires_grad_(True), "scales": pool["scales"][lo:hi].clone().requires_grad_(True),
                                           "rotq": pool["rotations"][lo:hi].clone().requires_grad_(True), "active_sh_degree": d}
                hu, sc_ = model(0, cut, deg), model(cut, P, 3)
                pkg = render_human_scene(data, hu, sc_, bg_color=st.bg, human_bg_color=torch.zeros(3, device=dev), render_mode="human_scene",
                                         render_human_separate=True)
                torch.autograd.backward([pkg["render"], pkg["human_img"]], [torch.full_like(pkg["render"], 1e-3), torch.full_like(pkg["render"], 2e-3)])
                outs.append((pkg["render"].detach(), pkg["human_img"].detach(), pkg["radii"], hu["xyz"].grad, hu["shs"].grad, sc_["xyz"].grad))
            gs_renderer._FRAME_CALL = True
            a, b = outs
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), f"render_pair differs on {(cut, P, H, W)}"
            for x, y in zip(a[3:], b[3:]):
                assert float((x - y).norm() / y.norm().clamp_min(1e-30)) <= 1e-4, f"render_pair gradients differ on {(cut, P, H, W)}"
            n_pair += 1
    n += 1
    if n % 64 == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
lib = dgr._load()
assert len(seen) > 256, f"only {len(seen)} distinct shapes: lengthen SOAK_SECONDS"
print(f"churn soak ok: {n} frames x 2 bindings (+{n_batch} deferred, +{n_pair} step pairs x 2 paths) in {time.time() - t0:.1f} s, {len(seen)} distinct (P, H, W) shapes, "
      f"{lib.hgs_debug_stat(b'tile_counter_entries')} counter arrays kept")
