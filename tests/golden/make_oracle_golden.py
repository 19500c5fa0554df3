#!/usr/bin/env python3
"""Writes tests/golden/oracle_cases.npz: inputs + every forward intermediate + all gradients of the fp32 C oracle
for two small seeded scenes.  The GPU tests compare the HIP path with these committed vectors too, so a silent
change of the oracle cannot move the goalposts; the CPU tests check the oracle still reproduces them.

    python tests/golden/make_oracle_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "ml-hugs_amd"), os.path.dirname(HERE)):
    sys.path.insert(0, p)

from oracle import hgs_oracle as ho  # noqa: E402
from scenes import make_scene, oracle_inputs  # noqa: E402

GOLDEN_CASES = {
    "g0": dict(P=120, H=48, W=64, seed=100, D=3, rotated_camera=True, wide=True),
    "g1": dict(P=90, H=40, W=40, seed=101, D=1, opaque=True, sigma_px=9.0),
}
FWD_KEYS = ("depths", "xy", "conic_opacity", "rgb", "cov3D", "clamped", "radii", "rect", "tiles_touched", "offsets",
            "keys", "values", "ranges", "color", "final_T", "n_contrib")
GRAD_KEYS = ("means3D", "means2D", "opacities", "shs", "scales", "rotations", "conic", "colors")


def main():
    ho.set_threads(1)  # a fixed summation order for the double accumulators
    out = {}
    for name, kw in GOLDEN_CASES.items():
        sc = make_scene(**kw)
        inp = oracle_inputs(sc)
        f = ho.forward(inp)
        g = ho.backward(inp, f, sc["dL_dpix"])
        for k in FWD_KEYS:
            out[f"{name}_fwd_{k}"] = f[k]
        out[f"{name}_fwd_N"] = np.int64(f["N"])
        for k in GRAD_KEYS:
            out[f"{name}_grad_{k}"] = g[k]
    path = os.path.join(HERE, "oracle_cases.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
