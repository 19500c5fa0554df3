#!/usr/bin/env python3
"""Where the fused tile-sort + forward-blend kernel's time goes, workgroup by workgroup (A/B builds with -DHGS_TRACE only:
`tools/ab_build.sh trace -DHGS_TRACE`, then `HGS_RASTERIZER_LIB=scratch/lib_trace.so HGS_BINDING=ctypes python tools/trace_fused.py`).
Every workgroup of the kernel stamps the 100 MHz wall clock at its start, after its sort (per-tile workgroups) / at the start
of its last quad (deep workers) and at its end; this prints the critical path of one C3 frame."""
import ctypes
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-hugs_amd"))
import diff_gaussian_rasterization as dgr                                # noqa: E402
from hugs_amd import synthetic as syn                                    # noqa: E402
from hugs_amd.renderer import render_human_scene                         # noqa: E402


def main_c2(trained=False):
    """the bench workload (200k / 1080p), or the trained-scene profile: residency and critical path of the fused kernel on a dense frame"""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    dev = torch.device("cuda:0")
    lib = dgr._load()
    P, H, W = 200_000, 1080, 1920
    cam = syn.pinhole_camera(H, W)
    g = syn.trained_scene_gaussians(P, cam, seed=0) if trained else syn.scene_gaussians(P, cam, seed=0, sigma_px=4.0)
    P = g["means3D"].shape[0]
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    t = {k: d(g[k]) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    st = GaussianRasterizationSettings(H, W, math.tan(cam["fovx"] * 0.5), math.tan(cam["fovy"] * 0.5), torch.ones(3, device=dev), 1.0,
                                       d(cam["world_view_transform"]), d(cam["full_proj_transform"]), 0 if trained else 3, d(cam["camera_center"]), False, False)
    run = lambda: GaussianRasterizer(st)(means3D=t["means3D"], means2D=torch.zeros(P, 3, device=dev), opacities=t["opacities"], shs=t["shs"],
                                         scales=t["scales"], rotations=t["rotations"])
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    nwg = 8160 + 24576
    buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    lib.hgs_debug_set_trace.argtypes = [ctypes.c_void_p]
    assert lib.hgs_debug_set_trace(buf.data_ptr()) == 0
    run()
    torch.cuda.synchronize()
    lib.hgs_debug_set_trace(None)
    report(buf.cpu().numpy().reshape(nwg, 8), nwg)


def main_joint(shell=False):
    """C4's joint human + scene render (110 210 + 200 000 Gaussians, 1080p): a dense frame whose body tiles are deep;
    shell: the human of tools/bench_step.py instead (a 1.2 m body SURFACE at 4 m: lists of several thousand entries)"""
    dev = torch.device("cuda:0")
    lib = dgr._load()
    H, W = 1080, 1920
    cam = syn.pinhole_camera(H, W)
    rng = np.random.default_rng(7)
    Ph, Ps = 110_210, 200_000
    q = rng.standard_normal((Ph, 4))
    hm = {"xyz": (rng.standard_normal((Ph, 3)) * np.array([0.22, 0.55, 0.14]) + np.array([0, 0, 4.0])).astype(np.float32),
          "scales": (0.035 / math.sqrt(Ph / 6890.0) * np.exp(0.3 * rng.standard_normal((Ph, 3)))).astype(np.float32),
          "rotq": (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32),
          "shs": (0.3 * rng.standard_normal((Ph, 16, 3))).astype(np.float32), "opacity": rng.uniform(0.05, 1.0, (Ph, 1)).astype(np.float32)}
    if shell:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from bench_knn import body_surface
        r = np.random.default_rng(3)
        hm["xyz"] = (body_surface(Ph, r, noise=0.004) * 0.6 + np.array([0, 0, 4.0])).astype(np.float32)
        hm["scales"] = (0.012 * np.exp(0.3 * r.standard_normal((Ph, 3)))).astype(np.float32)
    g = syn.scene_gaussians(Ps, cam, seed=8, sigma_px=4.0)
    sm = {"xyz": g["means3D"], "scales": g["scales"], "rotq": g["rotations"], "shs": g["shs"], "opacity": g["opacities"]}
    t = lambda a, grad=False: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev).requires_grad_(grad)
    human, scene = {k: t(v, True) for k, v in hm.items()}, {k: t(v, True) for k, v in sm.items()}
    human["active_sh_degree"], scene["active_sh_degree"] = 0, 3
    data = {k: (t(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    bg = torch.ones(3, device=dev)
    run = lambda: render_human_scene(data, human, scene, bg_color=bg, render_mode="human_scene")
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    nwg = 8160 + 24576
    buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    lib.hgs_debug_set_trace.argtypes = [ctypes.c_void_p]
    assert lib.hgs_debug_set_trace(buf.data_ptr()) == 0
    run()
    torch.cuda.synchronize()
    lib.hgs_debug_set_trace(None)
    report(buf.cpu().numpy().reshape(nwg, 8), nwg)


def main(P=110_210):
    dev = torch.device("cuda:0")
    lib = dgr._load()
    rng = np.random.default_rng(5)
    q = rng.standard_normal((P, 4))
    m = {"xyz": (rng.standard_normal((P, 3)) * np.array([0.22, 0.55, 0.14])).astype(np.float32),
         "scales": (0.035 / math.sqrt(P / 6890.0) * np.exp(0.3 * rng.standard_normal((P, 3)))).astype(np.float32),
         "rotq": (q / np.linalg.norm(q, axis=1, keepdims=True) * rng.uniform(0.8, 1.2, (P, 1))).astype(np.float32),
         "shs": (0.3 * rng.standard_normal((P, 16, 3))).astype(np.float32), "opacity": rng.uniform(0.05, 1.0, (P, 1)).astype(np.float32)}
    t = lambda a, grad=False: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev).requires_grad_(grad)
    human = {k: t(v, True) for k, v in m.items()}
    human["active_sh_degree"] = 0
    cam = syn.rotating_camera(3, 10, dist=5.0, fov=0.4, img_size=512)
    data = {k: (t(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    bg = torch.ones(3, device=dev)
    for _ in range(5):
        pkg = render_human_scene(data, human, None, bg_color=bg, render_mode="human")
    torch.cuda.synchronize()
    nwg = 2048 + 8 + 1024
    buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    lib.hgs_debug_set_trace.argtypes = [ctypes.c_void_p]
    assert lib.hgs_debug_set_trace(buf.data_ptr()) == 0
    pkg = render_human_scene(data, human, None, bg_color=bg, render_mode="human")
    torch.cuda.synchronize()
    lib.hgs_debug_set_trace(None)
    report(buf.cpu().numpy().reshape(nwg, 8), nwg)


def report(r, nwg):
    ran = r[:, 0] > 0
    t0 = r[ran, 0].min()
    us = lambda x: (x - t0) / 100.0
    end = np.where(r[:, 2] > 0, r[:, 2], r[:, 0])
    print(f"workgroups that ran: {ran.sum()}, kernel span (first start -> last end): {us(end[ran].max()):.1f} us")
    # residency: how many workgroups are on the chip at once, and where (HW_ID: cu_id bits 8..11, sh_id 12, se_id 13..15; XCC_ID bits 0..3)
    live = ran & (end > r[:, 0] + 100)   # lived longer than 1 us
    ev = sorted([(x, 1) for x in r[live, 0]] + [(x, -1) for x in end[live]])
    cur = peak = 0
    for _, d in ev:
        cur += d
        peak = max(peak, cur)
    xcc, hw = (r[:, 3] >> 32) & 0xF, r[:, 3] & 0xFFFFFFFF
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
    for t_us in (20, 50, 100):
        a = live & (r[:, 0] <= t0 + 100 * t_us) & (end > t0 + 100 * t_us)
        print(f"  resident at t = {t_us} us: {a.sum()}")
    at5 = live & (r[:, 0] <= t0 + 500) & (end > t0 + 500)
    per_cu = np.bincount(np.unique(cu[at5], return_inverse=True)[1]) if at5.any() else np.zeros(1, int)
    print(f"workgroups living > 1 us: {live.sum()}, peak resident {peak}; at t = 5 us: {at5.sum()} on {len(per_cu)} CUs (max {per_cu.max()} per CU, "
          f"histogram {np.bincount(per_cu).tolist()}); XCDs seen {sorted(set(xcc[ran].tolist()))}")
    workers = np.nonzero(ran & (r[:, 7] > 0))[0]
    tiles = np.nonzero(ran & (r[:, 7] == 0) & (r[:, 2] > 0))[0]
    print(f"deep workers with work: {len(workers)}, per-tile workgroups that blended: {len(tiles)}")
    if len(workers):
        w = r[workers]
        dur = (w[:, 2] - w[:, 1]) / 100.0
        print(f"  worker start (us): min {us(w[:, 0].min()):.1f} max {us(w[:, 0].max()):.1f};  items per worker: max {w[:, 7].max()}")
        print(f"  LAST item: duration us: mean {dur.mean():.1f} max {dur.max():.1f};  ends at: max {us(w[:, 2].max()):.1f}")
        k = np.argsort(-w[:, 2])[:8]
        for i in k:
            print(f"    wg {workers[i]}: start {us(w[i, 0]):.1f} last item {us(w[i, 1]):.1f} -> {us(w[i, 2]):.1f} us, n_quad {w[i, 4]}, rounds (all items) {w[i, 5]}, "
                  f"re-walks (wave 0) {w[i, 6]}, items {w[i, 7]}")
    if len(tiles):
        tl = r[tiles]
        print(f"  tile wg start: min {us(tl[:, 0].min()):.1f} max {us(tl[:, 0].max()):.1f}; sort us: mean {((tl[:, 1] - tl[:, 0]) / 100).mean():.1f} "
              f"max {((tl[:, 1] - tl[:, 0]) / 100).max():.1f}; blend us: mean {((tl[:, 2] - tl[:, 1]) / 100).mean():.1f} max {((tl[:, 2] - tl[:, 1]) / 100).max():.1f}")
        k = np.argsort(-tl[:, 2])[:8]
        for i in k:
            print(f"    wg {tiles[i]}: start {us(tl[i, 0]):.1f} sorted {us(tl[i, 1]):.1f} end {us(tl[i, 2]):.1f} us, n_tile {tl[i, 4] >> 32}, quad-0 list {tl[i, 4] & 0xFFFFFFFF}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] in ("joint", "step"):
        main_joint(sys.argv[1] == "step")
    elif len(sys.argv) > 1 and sys.argv[1] in ("c2", "trained"):
        main_c2(sys.argv[1] == "trained")
    else:
        main(int(sys.argv[1]) if len(sys.argv) > 1 else 110_210)
