#!/usr/bin/env python3
"""BASELINE config C3 as a timing: the HUGS human-only workload -- 110 210 Gaussians (SMPL, n_subdivision 2) at 512x512,
SH degree 0 on [P,16,3] storage, the canonical rotating-camera rig (dist 5, fov 0.4; gs_trainer.py:207-211), forward +
backward through the renderer adapter.  Prints one JSON line with per-stage times."""
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-hugs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from diff_gaussian_rasterization import profile_enable, profile_read   # noqa: E402
import bench_common as bc                                                # noqa: E402
from hugs_amd import synthetic as syn                                    # noqa: E402
from hugs_amd.renderer import render_human_scene                         # noqa: E402


def main(P=110_210, steps=200, warmup=20):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    q = rng.standard_normal((P, 4))
    m = {"xyz": (rng.standard_normal((P, 3)) * np.array([0.22, 0.55, 0.14])).astype(np.float32),
         "scales": (0.035 / math.sqrt(P / 6890.0) * np.exp(0.3 * rng.standard_normal((P, 3)))).astype(np.float32),
         "rotq": (q / np.linalg.norm(q, axis=1, keepdims=True) * rng.uniform(0.8, 1.2, (P, 1))).astype(np.float32),
         "shs": (0.3 * rng.standard_normal((P, 16, 3))).astype(np.float32), "opacity": rng.uniform(0.05, 1.0, (P, 1)).astype(np.float32)}
    t = lambda a, grad=False: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev).requires_grad_(grad)
    human = {k: t(v, True) for k, v in m.items()}
    human["active_sh_degree"] = 0
    cam = syn.rotating_camera(3, 10, dist=float(os.environ.get("HGS_C3_DIST", "5.0")), fov=0.4, img_size=512)   # (5.0: the reference's canonical rig)
    data = {k: (t(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    bg = torch.ones(3, device=dev)
    w = t(rng.standard_normal((3, 512, 512)) * 1e-3)
    leaves = [v for v in human.values() if isinstance(v, torch.Tensor)]

    def step():
        pkg = render_human_scene(data, human, None, bg_color=bg, render_mode="human")
        pkg["render"].backward(w)
        for x in leaves:
            x.grad = None
        return pkg

    for _ in range(warmup):
        pkg = step()
    ms, host_busy_us = bc.timed_loop(step, steps)
    profile_enable()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    stages = {k: round(v[0] / 10.0, 4) for k, v in profile_read().items()}
    profile_enable(())
    N, cap, has_long, sparse = bc.last_frame()
    Pv = int(pkg["visibility_filter"].sum())
    print(json.dumps({"workload": f"C3: {P} human Gaussians, 512x512, degree 0, rotating rig, fwd+bwd", "ms_per_step": round(ms, 4),
                      "fps": round(1e3 / ms, 1), "host_busy_us_per_step": round(host_busy_us, 1), "num_rendered_N": N, "binning_capacity": cap,
                      "sparse_frame": sparse, "has_long_tiles": has_long, "gaussians": P, "visible": Pv, "stages_ms": stages,
                      "roofline": bc.roofline(stages, P, Pv, N, 512, 512, 0),
                      "whole_frame": {"algorithmic_bytes": bc.frame_bytes(P, Pv, N, 512 * 512, 1024, 1),
                                      "GB_per_s": round(bc.frame_bytes(P, Pv, N, 512 * 512, 1024, 1) / (ms * 1e-3) / 1e9, 1)},
                      "box": bc.box()}))


if __name__ == "__main__":
    # HGS_BENCH_STEPS: short runs under the PMC passes of profiles/collect_workload.sh; an optional argument = P of the human
    kw = {}
    if os.environ.get("HGS_BENCH_STEPS"):
        kw = {"steps": int(os.environ["HGS_BENCH_STEPS"]), "warmup": 3}
    if len(sys.argv) > 1:
        kw["P"] = int(sys.argv[1])
    main(**kw)
