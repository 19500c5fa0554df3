#!/usr/bin/env python3
"""Row f-2 (second half) as a timing: the skinning step of the human model (lbs_extra's matmul / cat / bmm / slice +
the rotation product, hugs_trimlp.py:477-489,517) for the subdivided SMPL template's 110 210 Gaussians, forward +
backward: the reference's torch statements on the GPU vs the fused HIP path.  One JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-hugs_amd"))
from hugs_amd.lbs import lbs_skin   # noqa: E402


def torch_statements(A, W, v, R):
    """lbs.py:60-73 + hugs_trimlp.py:517 for one batch element, as the reference writes them"""
    T = torch.matmul(W[None], A.view(1, -1, 16)).view(1, -1, 4, 4)
    homo = torch.cat([v[None], torch.ones([1, v.shape[0], 1], dtype=v.dtype, device=v.device)], dim=2)
    verts = torch.matmul(T, homo.unsqueeze(-1))[:, :, :3, 0]
    return verts[0], T[0], T[0][:, :3, :3] @ R


def main(n=110_210, J=24, steps=200):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev).requires_grad_(True)
    A, W, v, R = mk(J, 4, 4), torch.softmax(4 * torch.randn(n, J, generator=g), -1).to(dev).requires_grad_(True), mk(n, 3), mk(n, 3, 3)
    gv, gT, gR = (torch.randn(*s, generator=g).to(dev) for s in ((n, 3), (n, 4, 4), (n, 3, 3)))
    res = {"workload": f"learned-LBS skinning of {n} Gaussians, {J} joints, forward + backward"}
    for name, fn in (("torch_statements", torch_statements), ("fused_hip", lbs_skin)):
        def step():
            verts, T, rot = fn(A, W, v, R)
            torch.autograd.backward([verts, T, rot], [gv, gT, gR])
            for x in (A, W, v, R):
                x.grad = None
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        res[name + "_us"] = round((time.perf_counter() - t0) / steps * 1e6, 1)
    res["speedup"] = round(res["torch_statements_us"] / res["fused_hip_us"], 2)
    res["bytes_per_gaussian_fwd_bwd"] = 4 * (J + 3 + 9 + 16 + 3 + 9) + 4 * (16 + J + 3 + 9 + 16 + 3 + 9 + 16 + J + 3 + 9 + 16 * 2 + J)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
