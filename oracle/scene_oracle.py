"""CPU oracle for row f-6 (test infrastructure only -- never imported by the product path): SceneGS.forward,
/root/reference/hugs/models/scene.py:147-160, restated in numpy float64 with its analytic backward.  Pinned by
tests/golden/reference_scene_forward.npz: outputs and autograd gradients of the reference's own `forward`, `get_features` and
`setup_functions`, compiled from its source file and run on CPU (tests/golden/make_golden_scene.py)."""
import numpy as np


def forward(scaling, rotation, opacity, features_dc, features_rest):
    s, r, o = (np.asarray(a, np.float64) for a in (scaling, rotation, opacity))
    n = np.maximum(np.linalg.norm(r, axis=1, keepdims=True), 1e-12)                  # F.normalize: x / max(|x|, eps)  (:50,149)
    shs = np.concatenate([np.asarray(features_dc, np.float64), np.asarray(features_rest, np.float64)], axis=1)   # :132-138
    return np.exp(s), r / n, 1.0 / (1.0 + np.exp(-o)), shs                           # :148,151 (torch.exp, torch.sigmoid)


def backward(scaling, rotation, opacity, g_scales, g_rotq, g_opacity, g_shs):
    s, r, o = (np.asarray(a, np.float64) for a in (scaling, rotation, opacity))
    n = np.linalg.norm(r, axis=1, keepdims=True)
    y = r / np.maximum(n, 1e-12)
    g = np.asarray(g_rotq, np.float64)
    d_rot = np.where(n > 1e-12, (g - y * (y * g).sum(1, keepdims=True)) / np.maximum(n, 1e-12), g / 1e-12)
    sig = 1.0 / (1.0 + np.exp(-o))
    g_shs = np.asarray(g_shs, np.float64)
    return np.asarray(g_scales, np.float64) * np.exp(s), d_rot, np.asarray(g_opacity, np.float64) * sig * (1 - sig), g_shs[:, :1], g_shs[:, 1:]
