"""Row f-6 -- SceneGS.forward fused (/root/reference/hugs/models/scene.py:147-160).
CPU: the numpy oracle against the outputs and autograd gradients of the reference's own methods
(tests/golden/make_golden_scene.py compiles them from /root/reference and runs them on CPU).
GPU: the HIP kernels through the drop-in Python function against the golden vectors and the oracle.  Tolerances (fp32):
values 2e-6 relative (expf / division: the last bit), gradients 1e-5 relative to the largest entry of the tensor."""
import os

import numpy as np
import pytest
import torch

from oracle import scene_oracle as so

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_scene_forward.npz"))
RAW = ("_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")
OUTS = ("scales", "rotq", "opacity", "shs")
VALUE_TOL, GRAD_TOL = 2e-6, 1e-5


def _close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-30)


def test_oracle_matches_the_reference_forward_and_backward():
    raw = [G[f"raw{k}"] for k in RAW]
    for name, got in zip(OUTS, so.forward(*raw)):
        assert _close(got, G[f"out_{name}"], VALUE_TOL), name
    grads = so.backward(raw[0], raw[1], raw[2], *(G[f"g_{k}"] for k in OUTS))
    for name, got in zip(RAW, grads):
        assert _close(got, G[f"grad{name}"], GRAD_TOL), name
    assert bytes(G["keys_json"]).decode().split(",") == ["xyz", "scales", "rotq", "shs", "opacity", "active_sh_degree"]


@pytest.fixture(scope="module")
def device():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_hip_matches_the_reference_vectors(device):
    from hugs_amd.scene_forward import scene_forward
    t = {k: torch.from_numpy(G[f"raw{k}"].copy()).to(device).requires_grad_(True) for k in RAW}
    xyz = torch.from_numpy(G["raw_xyz"]).to(device)
    out = scene_forward(xyz, *(t[k] for k in RAW), int(G["out_active_sh_degree"]))
    assert list(out.keys()) == bytes(G["keys_json"]).decode().split(",")
    assert out["xyz"] is xyz and out["active_sh_degree"] == int(G["out_active_sh_degree"])
    for name in OUTS:
        assert out[name].shape == G[f"out_{name}"].shape and _close(out[name].detach().cpu().numpy(), G[f"out_{name}"], VALUE_TOL), name
    assert np.array_equal(out["shs"].detach().cpu().numpy(), G["out_shs"])            # a copy: exact
    torch.autograd.backward([out[k] for k in OUTS], [torch.from_numpy(G[f"g_{k}"]).to(device) for k in OUTS])
    for name in RAW:
        assert _close(t[name].grad.cpu().numpy(), G[f"grad{name}"], GRAD_TOL), name
    assert np.array_equal(t["_features_rest"].grad.cpu().numpy(), G["grad_features_rest"])


@pytest.mark.gpu
@pytest.mark.parametrize("P,M", [(1, 16), (1000, 16), (4097, 4), (333, 1), (200_000, 16), (50, 9)])
def test_hip_against_the_oracle_and_partial_gradients(P, M, device):
    """Sizes around the block size, SH rows that are / are not whole float4s (M = 16, 4 / 9, 1: no higher coefficients), BASELINE's
    200 000; and a backward in which only some outputs received a gradient."""
    from hugs_amd.scene_forward import scene_activations
    r = np.random.default_rng(P + M)
    raw = [(r.standard_normal((P, 3)) - 2).astype(np.float32), r.standard_normal((P, 4)).astype(np.float32),
           r.standard_normal((P, 1)).astype(np.float32), r.standard_normal((P, 1, 3)).astype(np.float32),
           r.standard_normal((P, M - 1, 3)).astype(np.float32)]
    t = [torch.from_numpy(a.copy()).to(device).requires_grad_(True) for a in raw]
    outs = scene_activations(*t)
    for got, want in zip(outs, so.forward(*raw)):
        assert _close(got.detach().cpu().numpy(), want, VALUE_TOL)
    g = [r.standard_normal(tuple(o.shape)).astype(np.float32) for o in outs]
    torch.autograd.backward(list(outs), [torch.from_numpy(a).to(device) for a in g])
    for got, want in zip(t, so.backward(raw[0], raw[1], raw[2], *g)):
        assert got.grad.shape == want.shape and (want.size == 0 or _close(got.grad.cpu().numpy(), want, GRAD_TOL))
    t2 = [torch.from_numpy(a.copy()).to(device).requires_grad_(True) for a in raw]
    o2 = scene_activations(*t2)
    (o2[0].sum() + 2.0 * o2[2].sum()).backward()                                      # rotq and shs unused: no gradient for them
    assert t2[1].grad is None and t2[3].grad is None and t2[4].grad is None
    assert _close(t2[0].grad.cpu().numpy(), np.exp(raw[0].astype(np.float64)), GRAD_TOL)


@pytest.mark.gpu
def test_hip_scene_forward_feeds_the_rasterizer_like_the_torch_statements(device):
    """The fused dict through `render_human_scene(render_mode="scene")` equals the torch statements' dict through it (image and raw-parameter gradients)."""
    from hugs_amd import synthetic as syn
    from hugs_amd.renderer import render_human_scene
    from hugs_amd.scene_forward import scene_forward
    cam = syn.pinhole_camera(120, 160)
    g = syn.scene_gaussians(3000, cam, seed=3, sigma_px=3.0)
    data = {k: (torch.from_numpy(np.ascontiguousarray(v)).float().to(device) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
    raw_np = {"_xyz": g["means3D"], "_scaling": np.log(g["scales"]), "_rotation": g["rotations"] * 1.7,
              "_opacity": np.log(g["opacities"] / (1 - g["opacities"] + 1e-6) + 1e-6), "_features_dc": g["shs"][:, :1], "_features_rest": g["shs"][:, 1:]}
    w = torch.randn(3, 120, 160, device=device)
    results = []
    for fused in (True, False):
        t = {k: torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(device).requires_grad_(True) for k, v in raw_np.items()}
        if fused:
            d = scene_forward(t["_xyz"], t["_scaling"], t["_rotation"], t["_opacity"], t["_features_dc"], t["_features_rest"], 3)
        else:
            d = {"xyz": t["_xyz"], "scales": torch.exp(t["_scaling"]), "rotq": torch.nn.functional.normalize(t["_rotation"]),
                 "shs": torch.cat((t["_features_dc"], t["_features_rest"]), dim=1), "opacity": torch.sigmoid(t["_opacity"]), "active_sh_degree": 3}
        img = render_human_scene(data, None, d, bg_color=torch.ones(3, device=device), render_mode="scene")["render"]   # gs_trainer.py:264-272
        (img * w).sum().backward()
        results.append((img.detach(), {k: v.grad.clone() for k, v in t.items()}))
    (img_a, ga), (img_b, gb) = results
    assert (img_a - img_b).abs().max().item() <= 2e-6
    for k in ga:
        assert (ga[k] - gb[k]).abs().max().item() <= 1e-4 * max(gb[k].abs().max().item(), 1e-12), k
