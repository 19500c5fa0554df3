"""hugs_amd.spatial (CPU): the Morton permutation is a permutation, agrees between numpy and torch, keeps neighbours
together and puts non-finite positions last; permute_model re-indexes exactly the per-Gaussian tensors."""
import numpy as np
import torch

from hugs_amd.spatial import morton_order, permute_model


def test_morton_order_is_a_permutation_and_backend_independent():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((5000, 3)).astype(np.float32)
    a = morton_order(x)
    b = morton_order(torch.from_numpy(x)).numpy()
    assert sorted(a.tolist()) == list(range(5000))
    assert np.array_equal(a, b)


def test_morton_order_keeps_neighbours_together():
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (20000, 3)).astype(np.float32)
    o = morton_order(x)
    step_sorted = np.linalg.norm(np.diff(x[o], axis=0), axis=1).mean()
    step_random = np.linalg.norm(np.diff(x, axis=0), axis=1).mean()
    assert step_sorted < 0.2 * step_random


def test_non_finite_positions_go_last_and_empty_input_is_fine():
    x = np.random.default_rng(2).standard_normal((100, 3)).astype(np.float32)
    x[7, 1] = np.nan
    x[42, 0] = np.inf
    for o in (morton_order(x), morton_order(torch.from_numpy(x)).numpy()):
        assert set(o[-2:].tolist()) == {7, 42}
    assert len(morton_order(np.zeros((0, 3), np.float32))) == 0
    assert len(morton_order(torch.zeros(0, 3))) == 0


def test_permute_model_touches_only_per_gaussian_tensors():
    o = np.array([2, 0, 1])
    m = {"xyz": np.arange(9.0).reshape(3, 3), "shs": torch.arange(3 * 16 * 3.0).reshape(3, 16, 3), "active_sh_degree": 2,
         "bg": np.ones(4)}
    p = permute_model(m, o)
    assert np.array_equal(p["xyz"], m["xyz"][o]) and torch.equal(p["shs"], m["shs"][torch.from_numpy(o)])
    assert p["active_sh_degree"] == 2 and p["bg"] is m["bg"]
