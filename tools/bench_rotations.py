#!/usr/bin/env python3
"""Measurement for row f-7 on one MI355X: rotation_6d_to_matrix -> matrix_to_quaternion forward + backward for the 110 210 human
Gaussians of a HUGS step (hugs_trimlp.py:418-419), fused kernels against the reference's statements restated with torch ops (the
boolean-mask indexing and its two host synchronisations included), wall time per call.   python tools/bench_rotations.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-hugs_amd"))


def torch_m2q(m):     # the reference's statements restated with torch ops (boolean-mask indexing included)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(m.reshape(-1, 9), dim=-1)
    t = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1)
    qa = torch.zeros_like(t); pos = t > 0; qa[pos] = torch.sqrt(t[pos])
    rows = torch.stack([torch.stack([qa[:, 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1), torch.stack([m21 - m12, qa[:, 1] ** 2, m10 + m01, m02 + m20], -1),
                        torch.stack([m02 - m20, m10 + m01, qa[:, 2] ** 2, m12 + m21], -1), torch.stack([m10 - m01, m20 + m02, m21 + m12, qa[:, 3] ** 2], -1)], -2)
    cand = rows / (2.0 * qa[..., None].max(torch.tensor(0.1, device=m.device)))
    return cand[torch.nn.functional.one_hot(qa.argmax(-1), 4) > 0.5, :].reshape(-1, 4)


def torch_6d(d):
    a1, a2 = d[..., :3], d[..., 3:]
    b1 = torch.nn.functional.normalize(a1, dim=-1)
    b2 = torch.nn.functional.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), -2)


def main():
    from hugs_amd.rotations import matrix_to_quaternion, rotation_6d_to_matrix
    dev = torch.device("cuda:0"); n = 110_210
    d6 = torch.randn(n, 6, device=dev, requires_grad=True); g = torch.randn(n, 4, device=dev)
    def run(f6, fq):
        def step():
            d6.grad = None
            fq(f6(d6)).backward(g)
        for _ in range(5): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): step()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / 50 * 1e3
    print("6d -> matrix -> quaternion, fwd+bwd, 110210 rotations, ms per call (wall): fused %.4f, torch statements %.4f" % (run(rotation_6d_to_matrix, matrix_to_quaternion), run(torch_6d, torch_m2q)))


if __name__ == "__main__":
    main()
