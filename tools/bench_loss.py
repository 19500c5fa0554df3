#!/usr/bin/env python3
"""Measurement for row f-5 on one MI355X: the photometric loss of one training step (0.8 l1 + 0.2 (1 - ssim), forward and
backward, hugs/losses/loss.py:88-107) on a 1080p render, fused HIP kernels against the reference's torch statements run on
the same GPU, HIP events on the current stream.  Prints one JSON line.   python tools/bench_loss.py [--iters 50]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-hugs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def timed(fn, iters):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=int(os.environ.get("HGS_BENCH_STEPS", 50)))
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    a = ap.parse_args()
    from hugs_amd import losses
    from test_losses import _torch_statements
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    y = torch.rand(3, a.height, a.width, generator=g).to(dev)
    x = (y + 0.03 * torch.randn(y.shape, generator=g).to(dev)).clamp(0, 1).requires_grad_(True)

    def fused():
        x.grad = None
        (0.2 * (1.0 - losses.ssim(x, y)) + 0.8 * losses.l1_loss(x, y)).backward()

    def fused_forward():
        losses._LAST.clear()                      # (the pair's cached result would answer otherwise)
        with torch.no_grad():
            losses.l1_ssim(x, y)

    def statements():
        x.grad = None
        s, l1 = _torch_statements(x, y)
        (0.2 * (1.0 - s) + 0.8 * l1).backward()

    px = 3 * a.height * a.width
    out = {"workload": f"l1 + ssim, fwd+bwd, 3x{a.height}x{a.width} fp32", "fused_ms": round(timed(fused, a.iters), 4),
           "fused_forward_only_ms": round(timed(fused_forward, a.iters), 4), "torch_statements_ms": round(timed(statements, a.iters), 4),
           # algorithmic bytes: forward reads 2 images, writes 3 maps; backward reads 3 maps + 2 images, writes 1 gradient
           "algorithmic_bytes": 11 * 4 * px}
    out["fused_GBps"] = round(out["algorithmic_bytes"] / (out["fused_ms"] * 1e-3) / 1e9, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
