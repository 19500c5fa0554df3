"""CPU: the oracle checked against itself -- analytic backward (C) vs autograd through an independent dense torch
restatement (fp64), fp32 build vs fp64 build, and the committed oracle golden vectors."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import hgs_oracle as ho
from oracle import torch_oracle as to
from scenes import CASES, make_scene, oracle_inputs

SMALL = {
    "sh3_rot": dict(P=48, H=40, W=56, seed=0, D=3, rotated_camera=True),
    "sh2_wide_clamp": dict(P=40, H=48, W=48, seed=1, D=2, wide=True, sigma_px=10.0),
    "sh0_opaque": dict(P=60, H=32, W=32, seed=2, D=0, opaque=True, sigma_px=9.0),
    "rgb_mod": dict(P=30, H=33, W=47, seed=3, colors_precomp=True, scale_modifier=0.7, bg=(0.2, 0.5, 0.9)),
    "cov_precomp": dict(P=30, H=32, W=40, seed=4, D=1, cov3D_precomp=True),
}


def _torch_run(sc):
    t = lambda a: None if a is None else torch.tensor(np.asarray(a, np.float64), requires_grad=True)
    cam = sc["cam"]
    ins = {k: t(sc[k]) for k in ("means3D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp")}
    ins["means2D"] = torch.zeros(sc["means3D"].shape, dtype=torch.float64, requires_grad=True)
    c = lambda a: torch.tensor(np.asarray(a, np.float64))
    col, radii, aux = to.rasterize(ins["means3D"], ins["means2D"], ins["opacities"], c(cam["world_view_transform"]),
                                   c(cam["full_proj_transform"]), c(cam["camera_center"]), c(sc["bg"]), sc["tanfovx"],
                                   sc["tanfovy"], sc["H"], sc["W"], shs=ins["shs"], colors_precomp=ins["colors_precomp"],
                                   scales=ins["scales"], rotations=ins["rotations"], cov3D_precomp=ins["cov3D_precomp"],
                                   sh_degree=sc["D"], scale_modifier=sc["scale_modifier"])
    (col * c(sc["dL_dpix"])).sum().backward()
    return ins, col.detach().numpy(), radii.numpy(), aux


@pytest.mark.parametrize("name", list(SMALL))
def test_analytic_backward_equals_autograd_fp64(name):
    sc = make_scene(**SMALL[name])
    inp = oracle_inputs(sc, dtype=np.float64)
    f = ho.forward(inp)
    g = ho.backward(inp, f, sc["dL_dpix"])
    ins, col, radii, aux = _torch_run(sc)
    assert np.array_equal(radii, f["radii"])
    assert np.array_equal(aux["n_contrib"].numpy(), f["n_contrib"])
    np.testing.assert_allclose(col, f["color"], rtol=0, atol=1e-12)
    pairs = {"means3D": "means3D", "means2D": "means2D", "opacities": "opacities", "shs": "shs",
             "colors_precomp": "colors", "scales": "scales", "rotations": "rotations", "cov3D_precomp": "cov3D"}
    for k, gk in pairs.items():
        if ins.get(k) is None:
            continue
        a, b = ins[k].grad.numpy(), g[gk].reshape(ins[k].shape)
        denom = max(np.linalg.norm(a), 1e-30)
        # 1e-7 epsilons inside the analytic formulas (1/(den^2+1e-7)) bound the agreement at ~1e-10
        assert np.linalg.norm(a - b) / denom < 1e-9, (name, k, np.linalg.norm(a - b) / denom)


@pytest.mark.parametrize("name", list(SMALL))
def test_fp32_oracle_tracks_fp64_oracle(name):
    sc = make_scene(**SMALL[name])
    f32 = ho.forward(oracle_inputs(sc, np.float32))
    f64 = ho.forward(oracle_inputs(sc, np.float64))
    same = f32["radii"] == f64["radii"]
    assert same.mean() > 0.97  # ceil() of a rounded number may differ for a few
    if same.all() and np.array_equal(f32["n_contrib"], f64["n_contrib"]):
        np.testing.assert_allclose(f32["color"], f64["color"], atol=2e-5)
        g32 = ho.backward(oracle_inputs(sc, np.float32), f32, sc["dL_dpix"])
        g64 = ho.backward(oracle_inputs(sc, np.float64), f64, sc["dL_dpix"])
        for k in ("means3D", "means2D", "opacities", "scales", "rotations"):
            n = np.linalg.norm(g64[k])
            if n > 0:
                assert np.linalg.norm(g32[k] - g64[k]) / n < 2e-3, (name, k)


def test_oracle_reproduces_committed_golden_vectors():
    import importlib.util
    here = os.path.dirname(__file__)
    spec = importlib.util.spec_from_file_location("make_oracle_golden", os.path.join(here, "golden", "make_oracle_golden.py"))
    mog = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mog)
    gold = np.load(os.path.join(here, "golden", "oracle_cases.npz"))
    ho.set_threads(1)
    for name, kw in mog.GOLDEN_CASES.items():
        sc = make_scene(**kw)
        inp = oracle_inputs(sc)
        f = ho.forward(inp)
        g = ho.backward(inp, f, sc["dL_dpix"])
        for k in mog.FWD_KEYS:
            ref = gold[f"{name}_fwd_{k}"]
            if ref.dtype.kind in "iu":
                assert np.array_equal(f[k], ref), (name, k)
            else:
                np.testing.assert_allclose(f[k], ref, rtol=0, atol=1e-6, err_msg=f"{name} {k}")
        assert int(gold[f"{name}_fwd_N"]) == f["N"]
        for k in mog.GRAD_KEYS:
            np.testing.assert_allclose(g[k], gold[f"{name}_grad_{k}"], rtol=1e-5, atol=1e-6, err_msg=f"{name} grad {k}")


def test_sort_is_stable_and_ranges_partition_the_list():
    sc = make_scene(P=200, H=64, W=64, seed=30, D=0, sigma_px=12.0, with_culled=False)
    sc["means3D"][:, 2] = np.repeat(np.linspace(2, 6, 20), 10).astype(np.float32)  # many exact depth ties
    f = ho.forward(oracle_inputs(sc), stop_after="binning")
    keys, vals, rng = f["keys"], f["values"], f["ranges"]
    assert np.all(np.diff(keys.astype(np.uint64)) >= 0)
    tie = keys[1:] == keys[:-1]
    assert tie.any() and np.all(vals[1:][tie] > vals[:-1][tie])  # ties resolve by ascending Gaussian index
    covered = sum(int(e - s) for s, e in rng)
    assert covered == f["N"]
    for t, (s, e) in enumerate(rng):
        assert np.all((keys[s:e] >> np.uint64(32)) == t)


def test_empty_and_all_culled():
    sc = make_scene(P=0, H=16, W=16, seed=0, with_culled=False)
    f = ho.forward(oracle_inputs(sc))
    assert f["N"] == 0 and float(np.abs(f["color"]).max()) == 0.0  # zeros, not background
    sc = make_scene(P=20, H=16, W=16, seed=0, with_culled=False)
    sc["means3D"][:, 2] = 0.1
    f = ho.forward(oracle_inputs(sc))
    assert f["N"] == 0 and np.allclose(f["color"], 1.0)  # background only


# ---- a third, derivation-free check: central finite differences of the fp64 oracle FORWARD ---------------------------
# The analytic backward (oracle/hgs_oracle.c K7-K9, and the HIP K7-K9 typed from the same derivation) and the autograd
# restatement both encode somebody's reading of Appendix A.5; finite differences of the forward encode nothing.  The
# scenes avoid the three places where the reference's gradient is DELIBERATELY not the derivative (A.6: straight-through
# 0.99 cap -> opacities <= 0.9; frustum-clamp masks -> nothing outside 1.3 tanfov; culled Gaussians), so the plain
# derivative is the expected answer -- including the scale_modifier factor in dL/dscale.
FD_SCENES = {
    "sh3_rot_mod": dict(P=24, H=32, W=40, seed=21, D=3, rotated_camera=True, with_culled=False, scale_modifier=0.7, sigma_px=5.0),
    "rgb_cov_precomp": dict(P=20, H=32, W=32, seed=22, colors_precomp=True, cov3D_precomp=True, with_culled=False, sigma_px=5.0,
                            bg=(0.3, 0.6, 0.1)),
    "sh1_unitq": dict(P=32, H=24, W=48, seed=23, D=1, nonunit_quat=False, with_culled=False, sigma_px=4.0),
}


@pytest.mark.parametrize("name", list(FD_SCENES))
def test_analytic_backward_equals_finite_differences_of_the_forward_fp64(name):
    sc = make_scene(**FD_SCENES[name])
    sc["opacities"] = np.minimum(sc["opacities"], 0.9).astype(np.float32)   # alpha = o G never reaches the 0.99 cap
    ho.set_threads(1, np.float64)
    dL = sc["dL_dpix"].astype(np.float64)
    tensors = {"means3D": "means3D", "opacities": "opacities", "shs": "shs", "colors_precomp": "colors", "scales": "scales",
               "rotations": "rotations", "cov3D_precomp": "cov3D"}
    base = {k: None if sc[k] is None else np.asarray(sc[k], np.float64).copy() for k in tensors}

    def loss(vals):
        s = dict(sc)
        s.update(vals)
        f = ho.forward(oracle_inputs(s, dtype=np.float64))
        return float((f["color"] * dL).sum()), f

    l0, f0 = loss(base)
    assert (f0["radii"] > 0).all() and f0["N"] > sc["means3D"].shape[0]
    g = ho.backward(oracle_inputs(dict(sc, **base), dtype=np.float64), f0, dL)
    checked = 0
    for k, gk in tensors.items():
        if base[k] is None:
            continue
        ana = g[gk].reshape(base[k].shape)
        if k == "shs":   # coefficients above the active degree: no effect on the forward, zero gradient
            K = (sc["D"] + 1) ** 2
            assert float(np.abs(ana[:, K:]).max(initial=0.0)) == 0.0
        flat = base[k].reshape(-1)
        fd1, fd2 = np.zeros_like(flat), np.zeros_like(flat)
        idx = np.arange(flat.size) if k != "shs" else np.flatnonzero((np.arange(flat.size) // 3) % sc["M"] < (sc["D"] + 1) ** 2)
        for i in idx:
            h = 1e-6 * max(1.0, abs(flat[i]))
            for fd, step in ((fd1, h), (fd2, 2 * h)):
                v = flat.copy()
                v[i] += step
                lp, _ = loss({**base, k: v.reshape(base[k].shape)})
                v[i] -= 2 * step
                lm, _ = loss({**base, k: v.reshape(base[k].shape)})
                fd[i] = (lp - lm) / (2 * step)
        scale = np.abs(ana).max()
        # a step that moves some pixel's alpha across 1/255 (or T across 1e-4, or a radius across an integer) lands on a
        # jump of the forward: there the two step sizes disagree by ~2x and the element says nothing -- rare by construction
        smooth = np.abs(fd1 - fd2) <= 1e-5 * scale + 1e-7 * np.abs(fd1)
        usable = smooth[idx]
        assert usable.mean() >= 0.98, (name, k, float(usable.mean()))
        err = np.abs(fd1 - ana.reshape(-1))[idx][usable].max() / scale
        assert err <= 2e-6, (name, k, float(err))
        checked += int(usable.sum())
    assert checked >= 250


def test_upstream_scale_gradient_switch_of_the_oracle():
    """oracle_set_upstream_scale_grad: dL/dscale without the scale_modifier factor (the published kernel's convention; the
    library's HGS_BWD_UPSTREAM_SCALE_GRAD) = the true derivative / modifier, nothing else changes."""
    sc = make_scene(P=40, H=40, W=56, seed=17, D=1, scale_modifier=0.7, with_culled=False)
    inp = oracle_inputs(sc, np.float64)
    f = ho.forward(inp)
    g0 = ho.backward(inp, f, sc["dL_dpix"])
    try:
        ho.set_upstream_scale_grad(True)
        g1 = ho.backward(inp, f, sc["dL_dpix"])
    finally:
        ho.set_upstream_scale_grad(False)
    assert np.abs(g0["scales"]).max() > 0
    np.testing.assert_allclose(g1["scales"] * 0.7, g0["scales"], rtol=1e-12, atol=0)
    for k in ("means3D", "opacities", "shs", "rotations", "means2D"):
        assert np.array_equal(g0[k], g1[k]), k


def test_forward_does_not_depend_on_the_number_of_threads():
    """The per-Gaussian loops, key emission, the radix sort (chunked per thread, digit-major prefix) and the tile ranges run under
    OpenMP for the all-core CPU baseline (bench.py): every forward output is the serial one, bit for bit -- also on a list long
    enough for the sort to split into chunks (N >= 65 536)."""
    sc = make_scene(P=4000, H=96, W=128, seed=12, D=2, sigma_px=9.0)
    inp = oracle_inputs(sc)
    out = {}
    for n in (1, 7):
        ho.set_threads(n)
        out[n] = ho.forward(inp)
    # a team SMALLER than the number of chunks (what OMP_THREAD_LIMIT / OMP_DYNAMIC / a cgroup cap can hand the sort): no chunk is left out
    ho.set_threads(3)
    ho.set_sort_chunks(7)
    out["3 of 7"] = ho.forward(inp)
    ho.set_sort_chunks(0)
    ho.set_threads(1)
    assert out[1]["N"] >= 65536
    for k, a in out[1].items():
        assert np.array_equal(np.asarray(a), np.asarray(out[7][k])), k
        assert np.array_equal(np.asarray(a), np.asarray(out["3 of 7"][k])), k


def test_backward_does_not_depend_on_the_number_of_threads():
    """The pixel backward adds tile-local double sums into ONE shared double accumulator with atomics (round 4: the per-thread
    accumulators of round 3 capped the CPU baseline at 32 host threads): whatever the thread count and the order the tiles
    arrive in, the fp64 build agrees to ~1e-15 and the fp32 build -- rounded once from the double sums -- to the last bit but
    for a sum that lands on a rounding boundary."""
    sc = make_scene(**CASES["opaque_earlystop"])
    for dtype, tol in ((np.float64, 1e-13), (np.float32, 2e-7)):
        inp = oracle_inputs(sc, dtype) if dtype is np.float64 else oracle_inputs(sc)
        out = {}
        for n in (1, 8):
            ho.set_threads(n, dtype)
            f = ho.forward(inp)
            out[n] = ho.backward(inp, f, sc["dL_dpix"].astype(dtype))
        for k in out[1]:
            a, b = np.asarray(out[1][k], np.float64), np.asarray(out[8][k], np.float64)
            scale = max(np.abs(a).max(), 1e-300)
            assert np.abs(a - b).max() <= tol * scale, (dtype.__name__, k)
    ho.set_threads(1)
    ho.set_threads(1, np.float64)


def test_usable_cpus_respects_affinity_and_quota(tmp_path, monkeypatch):
    """The CPU baseline's thread count: never more than the affinity mask, and capped by a cgroup quota when one is set (the GPU
    boxes show 256 CPUs under a 16-CPU quota; 256 OpenMP threads ran the port 3.5x slower than 16)."""
    import builtins
    n = ho.usable_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            f = tmp_path / "cpu.max"
            f.write_text("300000 100000\n")
            return real_open(f, *a, **k)
        return real_open(path, *a, **k)

    monkeypatch.setattr(builtins, "open", fake_open)
    assert ho.usable_cpus() == min(3, len(os.sched_getaffinity(0)))

