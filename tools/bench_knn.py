#!/usr/bin/env python3
"""Measurement for SURVEY.md 8f row f-2 on one MI355X: smpl_lbsweight_top_k / knn_points at the shape the reference runs
every training step (hugs_trimlp.py:480-484: N_gs Gaussians x 6 890 SMPL vertices, K = 6), HIP events on the launch
stream.  Prints one JSON line.  The kernel is VALU-bound (n*m distance evaluations, 9 fp32 ops each); the figure of
merit is distance evaluations per second.   python tools/bench_knn.py [--points 110000] [--iters 20]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-hugs_amd"))


def body_surface(count, r, noise=0.0):
    """Points on the surface of a capsule figure of human size (torso, head, arms spread, legs): the shape of the problem the
    reference has -- SMPL's 6 890 vertices are a 2-D surface, and the Gaussians sit on or near it (hugs_trimlp.py:480)."""
    caps = [((0.0, 0.0, 0.0), (0.0, 0.55, 0.0), 0.14), ((0.0, 0.75, 0.0), (0.0, 0.78, 0.0), 0.10),
            ((0.18, 0.5, 0.0), (0.85, 0.5, 0.0), 0.045), ((-0.18, 0.5, 0.0), (-0.85, 0.5, 0.0), 0.045),
            ((0.09, -0.05, 0.0), (0.12, -0.9, 0.0), 0.07), ((-0.09, -0.05, 0.0), (-0.12, -0.9, 0.0), 0.07)]
    area = np.array([2 * np.pi * c[2] * (np.linalg.norm(np.subtract(c[1], c[0])) + 2 * c[2]) for c in caps])
    which = r.choice(len(caps), count, p=area / area.sum())
    out = np.empty((count, 3), np.float64)
    for k, (a, b, rad) in enumerate(caps):
        sel = np.nonzero(which == k)[0]
        a, b = np.asarray(a), np.asarray(b)
        axis = (b - a) / np.linalg.norm(b - a)
        length = np.linalg.norm(b - a)
        t = r.uniform(-rad, length + rad, len(sel))          # along the axis, the caps included
        d = r.standard_normal((len(sel), 3))
        d -= np.outer(d @ axis, axis)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        over = np.clip(t, 0, length) - t                      # inside a cap: tilt the radius towards the axis
        h = np.sqrt(np.maximum(rad * rad - over * over, 0.0))
        out[sel] = a + np.outer(np.clip(t, 0, length), axis) - np.outer(over, axis) + d * h[:, None]
    return (out + noise * r.standard_normal(out.shape)).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=110_000)
    ap.add_argument("--verts", type=int, default=6890)
    ap.add_argument("--iters", type=int, default=int(os.environ.get("HGS_BENCH_STEPS", 20)))
    ap.add_argument("--cpu-points", type=int, default=2000)
    ap.add_argument("--shape", choices=("body", "blob"), default="body",
                    help="body: a surface of human size (the reference's case); blob: a 3-D normal cloud with sparse tails")
    ap.add_argument("--no-grid", action="store_true", help="the scan of the whole template (no workspace)")
    a = ap.parse_args()
    from hugs_amd.knn import knn_points, smpl_lbsweight_top_k
    from oracle import knn_oracle as ko
    r = np.random.default_rng(0)
    if a.no_grid:
        os.environ["HGS_KNN_GRID"] = "0"
    if a.shape == "body":
        templ, pts = body_surface(a.verts, r), body_surface(a.points, r, noise=0.004)
    else:
        templ = (r.standard_normal((a.verts, 3)) * np.array([0.25, 0.6, 0.15])).astype(np.float32)
        pts = (templ[r.integers(0, a.verts, a.points)] + 0.02 * r.standard_normal((a.points, 3))).astype(np.float32)
    w = r.random((a.verts, 24)).astype(np.float32)
    w /= w.sum(1, keepdims=True)
    dev = torch.device("cuda:0")
    tp, tt, tw = torch.from_numpy(pts)[None].to(dev), torch.from_numpy(templ)[None].to(dev), torch.from_numpy(w).to(dev)
    out = {}
    for name, fn in (("smpl_lbsweight_top_k", lambda: smpl_lbsweight_top_k(tw, tp, tt)), ("knn_points", lambda: knn_points(tp, tt, K=6))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        out[name] = {"ms": round(ms, 4), "distance_evals_per_s": round(a.points * a.verts / (ms * 1e-3), 1)}
    t0 = time.perf_counter()
    ko.smpl_lbsweight_top_k(w, pts[:a.cpu_points], templ)
    cpu_s = time.perf_counter() - t0
    out["cpu_oracle"] = {"points": a.cpu_points, "s": round(cpu_s, 3),
                         "distance_evals_per_s": round(a.cpu_points * a.verts / cpu_s, 1), "kind": "port (numpy)"}
    print(json.dumps({"workload": f"{a.points} points x {a.verts} template vertices ({a.shape}), K=6, J=24",
                      "grid": not a.no_grid, **out}))


if __name__ == "__main__":
    main()
