#!/usr/bin/env python3
"""Generates tests/golden/reference_loss.npz: inputs, values and autograd gradients of the reference's own l1_loss and ssim
(/root/reference/hugs/losses/utils.py:54-108), whose function definitions are compiled from the reference's source file in
THIS container (read-only; the module itself imports pytorch3d, which is absent, so the five functions are taken one by one)
and run on CPU in fp32.  Only these vectors travel.

    python tests/golden/make_golden_loss.py

Cases: a smooth pair (a render against its target), pure noise, an image against itself, an image smaller than the window,
one whose sides are not multiples of the kernel's tile, and the masked l1 of hugs/losses/loss.py:89.
"""
import ast
import os
from math import exp

import numpy as np
import torch
import torch.nn.functional as F
from torch.autograd import Variable

REF = "/root/reference/hugs/losses/utils.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_loss.npz")


def main():
    ns = {"torch": torch, "F": F, "Variable": Variable, "exp": exp}
    tree = ast.parse(open(REF).read())
    for name in ("l1_loss", "gaussian", "create_window", "ssim", "_ssim"):
        node = next(f for f in tree.body if isinstance(f, ast.FunctionDef) and f.name == name)
        exec(compile(ast.Module(body=[node], type_ignores=[]), REF, "exec"), ns)
    ref_ssim, ref_l1 = ns["ssim"], ns["l1_loss"]
    r = np.random.default_rng(5)

    def smooth(c, h, w):
        yy, xx = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing="ij")
        base = np.stack([0.5 + 0.4 * np.sin(6.0 * xx + k) * np.cos(4.0 * yy - k) for k in range(c)])
        return base.astype(np.float32)

    cases = {}
    a = smooth(3, 37, 53)
    cases["smooth"] = (np.clip(a + 0.05 * r.standard_normal(a.shape), 0, 1).astype(np.float32), a)
    cases["noise"] = (r.random((3, 24, 31)).astype(np.float32), r.random((3, 24, 31)).astype(np.float32))
    b = r.random((1, 19, 70)).astype(np.float32)
    cases["same"] = (b.copy(), b)
    cases["tiny"] = (r.random((3, 5, 7)).astype(np.float32), r.random((3, 5, 7)).astype(np.float32))
    c = smooth(3, 40, 130)
    cases["ragged"] = ((c + 0.1 * r.standard_normal(c.shape)).astype(np.float32), c)      # (values outside [0,1] too)
    out = {"window_1d": ns["gaussian"](11, 1.5).numpy(), "window_2d": ns["create_window"](11, 1)[0, 0].numpy()}
    for name, (x, y) in cases.items():
        tx = torch.from_numpy(x.copy()).requires_grad_(True)
        ty = torch.from_numpy(y)
        s = ref_ssim(tx, ty)
        l1 = ref_l1(tx, ty)
        (0.2 * (1.0 - s) + 0.8 * l1).backward()                      # the weights of hugs/losses/loss.py:19-20 (l_ssim_w 0.2, l_l1_w 0.8)
        out.update({f"{name}_x": x, f"{name}_y": y, f"{name}_ssim": np.float32(s.item()), f"{name}_l1": np.float32(l1.item()),
                    f"{name}_grad": tx.grad.numpy().copy()})
        tx.grad = None
        ref_ssim(tx, ty).backward()
        out[f"{name}_grad_ssim"] = tx.grad.numpy().copy()
    x, y = cases["smooth"]
    mask = (r.random((1, 37, 53)) > 0.6).astype(np.float32)
    tx = torch.from_numpy(x.copy()).requires_grad_(True)
    lm = ref_l1(tx, torch.from_numpy(y), torch.from_numpy(mask))
    lm.backward()
    out.update(masked_mask=mask, masked_l1=np.float32(lm.item()), masked_grad=tx.grad.numpy().copy())
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
