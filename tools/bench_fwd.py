#!/usr/bin/env python3
"""Forward-only frame loops (the reference's validate / animate / render_canonical, gs_trainer.py:448-684): frames per
second of render() called frame by frame under no_grad vs hugs_amd.renderer.render_batch (deferred frames, side streams),
at 1080p for several Gaussian counts.  One JSON line per count."""
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-hugs_amd"))
from hugs_amd import synthetic as syn                       # noqa: E402
from hugs_amd.renderer import render, render_batch          # noqa: E402


def main(counts=(50_000, 100_000, 200_000), H=1080, W=1920, D=3, frames=240):
    dev = torch.device("cuda:0")
    cam0 = syn.pinhole_camera(H, W)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for P in counts:
        g = syn.scene_gaussians(P, cam0, seed=0, sigma_px=4.0)
        G = {k: t(v) for k, v in g.items()}
        fl = []
        for i in range(frames):
            yaw = math.radians(0.05) * (i - frames // 2)
            w2c = np.eye(4)
            w2c[0, 0], w2c[0, 2], w2c[2, 0], w2c[2, 2] = math.cos(yaw), math.sin(yaw), -math.sin(yaw), math.cos(yaw)
            cam = syn.camera_from_w2c(w2c, cam0["fovx"], cam0["fovy"], H, W)
            data = {k: (t(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
            fl.append(dict(means3D=G["means3D"], feats=G["shs"], opacity=G["opacities"], scales=G["scales"], rotations=G["rotations"],
                           data=data, bg_color=torch.ones(3, device=dev), active_sh_degree=D))
        res = {"workload": f"forward only, {P} Gaussians, {W}x{H}, SH degree {D}, {frames} cameras"}
        with torch.no_grad():
            for _ in range(2):
                [render(**fr) for fr in fl[:20]]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for fr in fl:
                render(**fr)
            torch.cuda.synchronize()
            res["serial_render_fps"] = round(frames / (time.perf_counter() - t0), 1)
        # chunks of 24 frames, as a validation loop would consume them (the images of a chunk are dropped before the
        # next one is rendered, so the caching allocator recycles their blocks); second pass timed
        for ns in (1, 2, 3):
            for timed in (False, True):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for c in range(0, frames, 24):
                    out = render_batch(fl[c:c + 24], num_streams=ns)
                    del out
                torch.cuda.synchronize()
                if timed:
                    res[f"render_batch_{ns}_streams_fps"] = round(frames / (time.perf_counter() - t0), 1)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
