#!/usr/bin/env python3
"""BASELINE config C4 as a timing: one HUGS training step's rasterizer work -- the joint human+scene render (110 210 +
200 000 Gaussians, 1080p, SH degree 0 taken from the human model) and the separate human-only render, one backward
through both (gs_renderer.py:56,69; hugs_human_scene.yaml humansep_w).  Prints one JSON line.
HGS_CONCURRENT_RENDERS=0/1 (read at import) selects whether the second render runs on a side stream."""
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
import bench_common as bc                                  # noqa: E402
from hugs_amd import synthetic as syn                      # noqa: E402
from hugs_amd.renderer import render_human_scene           # noqa: E402


def main(steps=100, warmup=15):
    dev = torch.device("cuda:0")
    H, W = (int(v) for v in os.environ.get("HGS_C4_SIZE", "1080x1920").split("x"))   # (HxW; the BASELINE config is 1080p)
    cam = syn.pinhole_camera(H, W)
    rng = np.random.default_rng(7)
    Ph, Ps = 110_210, 200_000
    hm = {"xyz": (rng.standard_normal((Ph, 3)) * np.array([0.22, 0.55, 0.14]) + np.array([0, 0, 4.0])).astype(np.float32),
          "scales": (0.035 / math.sqrt(Ph / 6890.0) * np.exp(0.3 * rng.standard_normal((Ph, 3)))).astype(np.float32),
          "rotq": None, "shs": (0.3 * rng.standard_normal((Ph, 16, 3))).astype(np.float32),
          "opacity": rng.uniform(0.05, 1.0, (Ph, 1)).astype(np.float32)}
    q = rng.standard_normal((Ph, 4))
    hm["rotq"] = (q / np.linalg.norm(q, axis=1, keepdims=True) * rng.uniform(0.8, 1.2, (Ph, 1))).astype(np.float32)  # non-unit, as HUGS feeds
    g = syn.scene_gaussians(Ps, cam, seed=8, sigma_px=4.0)
    sm = {"xyz": g["means3D"], "scales": g["scales"], "rotq": g["rotations"], "shs": g["shs"], "opacity": g["opacities"]}
    t = lambda a, grad=False: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev).requires_grad_(grad)
    human = {k: t(v, True) for k, v in hm.items()}
    scene = {k: t(v, True) for k, v in sm.items()}
    human["active_sh_degree"], scene["active_sh_degree"] = 0, 3
    data = {k: (t(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    bg, hbg = torch.ones(3, device=dev), torch.zeros(3, device=dev)
    w1 = t(rng.standard_normal((3, H, W)) * 1e-3)
    w2 = t(rng.standard_normal((3, H, W)) * 1e-3)
    leaves = [v for m in (human, scene) for v in m.values() if isinstance(v, torch.Tensor)]

    def step():
        pkg = render_human_scene(data, human, scene, bg_color=bg, human_bg_color=hbg, render_mode="human_scene",
                                 render_human_separate=True)
        # fixed dL/dimage for both renders, handed to autograd directly (as bench.py does): no stand-in loss kernels in the timing
        torch.autograd.backward([pkg["render"], pkg["human_img"]], [w1, w2])
        for x in leaves:
            x.grad = None

    for _ in range(warmup):
        step()
    torch.cuda.reset_peak_memory_stats(dev)
    ms, host_busy_us = bc.timed_loop(step, steps)
    extra = {"host_busy_us_per_step": round(host_busy_us, 1), "peak_memory_MB": round(torch.cuda.max_memory_allocated(dev) / 2**20, 1)}
    with torch.no_grad():   # the two frames' exact integers (N of the joint frame = this thread's last forward on the statement path too)
        pk = render_human_scene(data, human, scene, bg_color=bg, human_bg_color=hbg, render_mode="human_scene", render_human_separate=False)
        extra["joint_frame"] = {"gaussians": Ph + Ps, "visible": int(pk["visibility_filter"].sum()), "num_rendered_N": bc.last_frame()[0]}
        pk = render_human_scene(data, human, None, bg_color=hbg, render_mode="human")
        extra["human_only_frame"] = {"gaussians": Ph, "visible": int(pk["visibility_filter"].sum()), "num_rendered_N": bc.last_frame()[0]}
    if os.environ.get("HGS_C4_WITH_LOSS", "0") == "1":
        # the same step with the reference's photometric loss on both renders as the source of dL/dimage
        # (hugs/losses/loss.py:88-107,128-137: 0.8 l1 + 0.2 (1 - ssim)): fused row f-5 kernels, then the torch statements
        from hugs_amd import losses
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        sys.path.insert(0, ROOT)
        from test_losses import _torch_statements
        gt1, gt2 = torch.rand(3, H, W, device=dev), torch.rand(3, H, W, device=dev)

        def fused_loss(a, b):
            return 0.8 * losses.l1_loss(a, b) + 0.2 * (1.0 - losses.ssim(a, b))

        def torch_loss(a, b):
            s_, l1_ = _torch_statements(a, b)
            return 0.8 * l1_ + 0.2 * (1.0 - s_)

        for name, fn, n in (("fused_loss", fused_loss, steps), ("torch_loss", torch_loss, max(steps // 5, 3))):
            def step_l():
                pkg = render_human_scene(data, human, scene, bg_color=bg, human_bg_color=hbg, render_mode="human_scene",
                                         render_human_separate=True)
                (fn(pkg["render"], gt1) + fn(pkg["human_img"], gt2)).backward()
                for x in leaves:
                    x.grad = None
            for _ in range(3):
                step_l()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                step_l()
            torch.cuda.synchronize()
            extra[f"ms_per_training_step_raster_and_{name}"] = round((time.perf_counter() - t0) / n * 1e3, 4)
    from diff_gaussian_rasterization import profile_enable, profile_read
    profile_enable()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    stages = {k: round(v[0] / 5.0, 4) for k, v in profile_read().items()}
    profile_enable(())
    jf, hf = extra["joint_frame"], extra["human_only_frame"]
    T = ((H + 15) // 16) * ((W + 15) // 16)
    # both renders' launches of a stage are summed in stages_ms: so are their algorithmic bytes (degree 0: K = 1)
    sb = {k: a + b for (k, a), b in zip(bc.stage_bytes(Ph + Ps, jf["visible"], jf["num_rendered_N"], H * W, T, 1).items(),
                                        bc.stage_bytes(Ph, hf["visible"], hf["num_rendered_N"], H * W, T, 1).values())}
    dom = max((k for k in stages if k in sb), key=lambda k: stages[k])
    gbps = sb[dom] / (stages[dom] * 1e-3) / 1e9
    extra["roofline"] = {"bound": "hbm", "kernel": bc.kernel_of(dom, sparse=False, with_checkpoints=False) if dom == "blend_backward" else bc.kernel_of(dom), "stage": dom, "achieved": round(gbps, 1), "peak": bc.HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(gbps / bc.HBM_PEAK_GBPS, 5), "algorithmic_bytes_both_renders": int(sb[dom]), "stage_ms_both_renders": stages[dom],
                         "kernel_note": "both renders' launches of the stage are summed; the name is the JOINT render's kernel (the human-only render's blend backward is "
                                        "blend_backward_segmented_kernel)" if dom == "blend_backward" else None,
                         "traffic": None}
    extra["box"] = bc.box()
    print(json.dumps({"stages_ms_both_renders": stages, "workload": f"C4: joint (110210+200000) + human-only renders, {W}x{H}, fwd+bwd through both",
                      "concurrent_renders": os.environ.get("HGS_CONCURRENT_RENDERS", "1") != "0",
                      "joint_render": "torch.cat (reference form)" if os.environ.get("HGS_JOINT_CONCAT", "0") == "1" else "second segment (no concatenation)",
                      "ms_per_training_step_raster": round(ms, 4), "steps_per_s": round(1e3 / ms, 1), **extra}))


if __name__ == "__main__":
    # HGS_BENCH_STEPS: short runs under the PMC passes of profiles/collect_workload.sh; an optional argument = P of the human
    kw = {}
    if os.environ.get("HGS_BENCH_STEPS"):
        kw = {"steps": int(os.environ["HGS_BENCH_STEPS"]), "warmup": 3}
    main(**kw)
