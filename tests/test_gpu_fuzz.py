"""Randomised parity: scenes drawn from the whole parameter space of tests/scenes.py (Gaussian count, ragged image sizes,
SH degree, splat size over two decades, focal length, background, quaternion norm, camera pose, precomputed colour /
covariance, scale modifier, opaque / off-frustum / culled populations) through the drop-in API against the oracle, with
the bars of test_gpu_parity.py: radii and N exact, image within tolerance, every gradient within 1e-3 relative L2.
One allowance on top (about one scene in a thousand needs it): where the image shows a pixel that took the other branch of
a threshold (alpha within rounding of 1/255, T of 1e-4 -- the GPU's exp2-domain alpha and the oracle's differ by ~1e-6
relative), the Gaussians behind that pixel carry a finite gradient jump; a tensor is then also accepted if it is within
the bar once its three worst Gaussians are set aside.
HGS_FUZZ_SCENES (default 48) scenes from seed HGS_FUZZ_SEED (default 0); a failure names the scene's kwargs."""
import os

import numpy as np
import pytest
import torch

from oracle import hgs_oracle as ho
from scenes import make_scene, oracle_inputs
from test_gpu_parity import COLOR_TOL, GRAD_REL_TOL, _stacked_scene, check_image, rel_l2, run_gpu, to_dev

pytestmark = pytest.mark.gpu


def draw(rng):
    kw = dict(P=int(rng.integers(1, 900)), H=int(rng.integers(17, 180)), W=int(rng.integers(17, 220)),
              seed=int(rng.integers(0, 1 << 20)), D=int(rng.integers(0, 4)),
              sigma_px=float(np.exp(rng.uniform(np.log(0.6), np.log(60.0)))), focal_frac=float(rng.uniform(0.35, 1.2)),
              bg=tuple(float(v) for v in rng.uniform(0, 1, 3)), nonunit_quat=bool(rng.integers(0, 2)),
              rotated_camera=bool(rng.integers(0, 2)), scale_modifier=float(rng.choice([1.0, 1.0, 0.4, 1.7])),
              opaque=rng.random() < 0.2, wide=rng.random() < 0.25, with_culled=rng.random() < 0.7)
    mode = rng.random()
    if mode < 0.15:
        kw["colors_precomp"] = True
    elif mode < 0.3:
        kw["cov3D_precomp"] = True
    return kw


def test_random_scenes_match_the_oracle(device):
    import diff_gaussian_rasterization as dgr
    rng = np.random.default_rng(int(os.environ.get("HGS_FUZZ_SEED", "0")))
    n_scenes = int(os.environ.get("HGS_FUZZ_SCENES", "48"))
    for k in range(n_scenes):
        if rng.random() < 0.06:   # a pile of small Gaussians over a few tiles: lists of 1k..10k entries (the long-list sort paths)
            kw = dict(P=int(rng.integers(1200, 9000)), H=int(rng.integers(40, 150)), W=int(rng.integers(40, 150)),
                      seed=int(rng.integers(0, 1 << 20)), spread_px=float(rng.uniform(3.0, 20.0)))
            sc = _stacked_scene(**kw)
        else:
            kw = draw(rng)
            sc = make_scene(**kw)
        what = f"scene {k}: {kw}"
        inp = oracle_inputs(sc)
        ref = ho.forward(inp)
        ref_g = ho.backward(inp, ref, sc["dL_dpix"])
        _, first_color, _ = run_gpu(sc, device)    # a shape's first frame waits for N before it bins; ...
        t, color, radii = run_gpu(sc, device)       # ... the next one is enqueued whole on a guess of N
        assert torch.equal(first_color, color), what
        assert np.array_equal(radii.cpu().numpy(), ref["radii"]), what
        assert dgr.last_frame_info()[0] == ref["N"], what
        img = color.detach().cpu().numpy()
        check_image(img, ref["color"], what)
        flipped_pixels = int((np.abs(img - ref["color"]) > COLOR_TOL).any(axis=0).sum())
        color.backward(to_dev(sc["dL_dpix"], device))
        torch.cuda.synchronize()
        pairs = [("means3D", "means3D"), ("means2D", "means2D"), ("opacities", "opacities")]
        pairs.append(("shs", "shs") if sc["shs"] is not None else ("colors_precomp", "colors"))
        pairs += [("scales", "scales"), ("rotations", "rotations")] if sc["cov3D_precomp"] is None else [("cov3D_precomp", "cov3D")]
        for name, ref_name in pairs:
            g = t[name].grad.cpu().numpy()
            r = ref_g[ref_name]
            assert np.isfinite(g).all(), f"{what}: non-finite gradient in {name}"
            # (a tensor whose reference gradient is ~0 -- e.g. everything culled -- is compared absolutely)
            g = g.reshape(r.shape)
            err = rel_l2(g, r)
            if err > GRAD_REL_TOL and flipped_pixels:
                worst = np.argsort(-np.abs(g - r).reshape(g.shape[0], -1).max(axis=1))[:3]
                keep = np.ones(g.shape[0], bool)
                keep[worst] = False
                err = rel_l2(g[keep], r[keep])
            assert err <= GRAD_REL_TOL or np.abs(g - r).max() <= 1e-7, f"{what}: grad {name} rel L2 err {err:.3e}"
