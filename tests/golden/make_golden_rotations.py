#!/usr/bin/env python3
"""Generates tests/golden/reference_rotations.npz: inputs, outputs and autograd gradients of the reference's own
rotation_6d_to_matrix, _sqrt_positive_part and matrix_to_quaternion (/root/reference/hugs/utils/rotations.py), compiled from
the source file in THIS container (read-only) and run on CPU in fp32.  Only these vectors travel.
    python tests/golden/make_golden_rotations.py
Cases: random rotations (all four candidate rows occur), rotations by exactly pi (a trace term of exactly 0: the zero
subgradient; q_abs below the 0.1 floor for the others), scaled / sheared matrices as the LBS blend of hugs_trimlp.py:517
produces (not orthonormal), 6-D inputs incl. a zero first vector and two parallel vectors (normalize's eps branches)."""
import ast
import os

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference/hugs/utils/rotations.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_rotations.npz")


def main():
    ns = {"torch": torch, "F": F}
    tree = ast.parse(open(REF).read())
    for name in ("_sqrt_positive_part", "matrix_to_quaternion", "rotation_6d_to_matrix"):
        node = next(f for f in tree.body if isinstance(f, ast.FunctionDef) and f.name == name)
        exec(compile(ast.Module(body=[node], type_ignores=[]), REF, "exec"), ns)
    r = np.random.default_rng(31)
    d6 = r.standard_normal((300, 6)).astype(np.float32)
    d6[0, :3] = 0.0                                                   # zero first vector
    d6[1, 3:] = 2.0 * d6[1, :3]                                       # parallel vectors: u = 0
    t6 = torch.from_numpy(d6.copy()).requires_grad_(True)
    R = ns["rotation_6d_to_matrix"](t6)
    gR = r.standard_normal((300, 3, 3)).astype(np.float32)
    R.backward(torch.from_numpy(gR))
    out = {"d6": d6, "d6_matrix": R.detach().numpy(), "d6_g": gR, "d6_grad": t6.grad.numpy().copy()}
    mats = R.detach().numpy()[2:].copy()                              # proper rotations
    pi_rots = np.stack([np.diag(v).astype(np.float32) for v in ([1, -1, -1], [-1, 1, -1], [-1, -1, 1], [1, 1, 1])])
    blended = (mats[:60] * r.uniform(0.6, 1.3, (60, 1, 1)) + 0.05 * r.standard_normal((60, 3, 3))).astype(np.float32)
    M = np.concatenate([mats, pi_rots, blended]).astype(np.float32)
    tM = torch.from_numpy(M.copy()).requires_grad_(True)
    q = ns["matrix_to_quaternion"](tM)
    gq = r.standard_normal(tuple(q.shape)).astype(np.float32)
    q.backward(torch.from_numpy(gq))
    out.update(matrix=M, quat=q.detach().numpy(), quat_g=gq, matrix_grad=tM.grad.numpy().copy())
    np.savez_compressed(OUT, **out)
    rows = np.bincount(np.abs(out["quat"]).argmax(1), minlength=4)
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT) / 1024:.1f} KiB; largest component per case: {rows}")


if __name__ == "__main__":
    main()
