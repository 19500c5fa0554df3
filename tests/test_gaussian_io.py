"""SURVEY.md 8f row f-3 -- the on-disk Gaussian format (PLY) and camera construction, host-side.
The column order is pinned by the reference's own `construct_list_of_attributes` (executed from /root/reference by
tests/golden/make_golden.py); the reference's writer/reader go through the `plyfile` package, which is not installed, so
the byte layout is pinned by the PLY specification (binary_little_endian, float properties) and by round trips."""
import json
import math
import os

import numpy as np
import pytest
import torch

from hugs_amd import gaussian_io as gio
from hugs_amd import synthetic as syn

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_substeps.npz"))


def params(P=37, K=16, seed=0):
    g = torch.Generator().manual_seed(seed)
    return {"xyz": torch.randn(P, 3, generator=g), "features_dc": torch.randn(P, 1, 3, generator=g),
            "features_rest": torch.randn(P, K - 1, 3, generator=g), "opacity": torch.randn(P, 1, generator=g),
            "scaling": torch.randn(P, 3, generator=g) - 3.0, "rotation": torch.randn(P, 4, generator=g)}


def test_attribute_order_is_the_reference_s():
    assert gio.attribute_names() == json.loads(bytes(G["ply_attribute_names_json"]).decode())


def test_ply_round_trip_and_layout(tmp_path):
    p = params()
    path = str(tmp_path / "sub" / "point_cloud.ply")
    gio.write_gaussian_ply(path, **p)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 37\nproperty float x\n")
    assert len(body) == 37 * 62 * 4                                  # 3+3+3+45+1+3+4 float columns
    rows = np.frombuffer(body, "<f4").reshape(37, 62)
    assert np.array_equal(rows[:, 3:6], np.zeros((37, 3), np.float32))           # normals
    # channel-major flattening (transpose(1,2).flatten): f_rest_0..14 are the RED coefficients 1..15
    assert np.array_equal(rows[:, 9:9 + 15], p["features_rest"][:, :, 0].numpy())
    assert np.array_equal(rows[:, 9 + 15:9 + 30], p["features_rest"][:, :, 1].numpy())
    back = gio.read_gaussian_ply(path, max_sh_degree=3)
    for k, v in p.items():
        assert back[k].shape == v.shape and torch.equal(back[k], v), k
    with pytest.raises(ValueError):
        gio.read_gaussian_ply(path, max_sh_degree=2)                  # f_rest_* count must match the degree
    act = gio.activated(back)
    assert act["shs"].shape == (37, 16, 3) and torch.equal(act["shs"][:, 0], p["features_dc"][:, 0])
    assert torch.allclose(act["rotq"].norm(dim=-1), torch.ones(37)) and (act["scales"] > 0).all()
    assert torch.equal(act["opacity"], torch.sigmoid(p["opacity"]))


def test_reads_ascii_and_shuffled_property_order(tmp_path):
    names = gio.attribute_names(3, 0, 3, 4)
    shuffled = names[::-1]
    vals = {n: [float(i) + 0.25 * k for k in range(2)] for i, n in enumerate(names)}
    path = str(tmp_path / "a.ply")
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\n" +
                "".join(f"property float {n}\n" for n in shuffled) + "element face 0\nproperty list uchar int vertex_indices\nend_header\n")
        for k in range(2):
            f.write(" ".join(repr(vals[n][k]) for n in shuffled) + "\n")
    d = gio.read_gaussian_ply(path, max_sh_degree=0)
    assert d["xyz"].tolist() == [[0.0, 1.0, 2.0], [0.25, 1.25, 2.25]]
    assert d["features_rest"].shape == (2, 0, 3) and d["rotation"].shape == (2, 4)
    assert d["rotation"][1].tolist() == [vals[f"rot_{i}"][1] for i in range(4)]


def test_camera_from_colmap_matches_reference_conventions():
    for (fx, fy), ref in zip(G["proj_fovs"], G["proj_mats"]):         # the reference's get_projection_matrix
        assert np.array_equal(gio.projection_matrix(0.01, 100.0, fx, fy).numpy(), ref)
    H, W, f = 480, 640, 500.0
    K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]])
    yaw = 0.3
    R = np.array([[math.cos(yaw), 0, math.sin(yaw)], [0, 1, 0], [-math.sin(yaw), 0, math.cos(yaw)]])
    t = np.array([0.1, -0.2, 3.0])
    w2c = np.eye(4, dtype=np.float32)
    w2c[:3, :3], w2c[:3, 3] = R, t
    cam = gio.camera_from_colmap(K, w2c, H, W)
    assert cam["image_height"] == H and cam["image_width"] == W
    assert abs(cam["fovx"] - 2 * math.atan(W / (2 * f))) < 1e-12
    assert torch.equal(cam["world_view_transform"], torch.from_numpy(w2c).T)
    np.testing.assert_allclose(cam["camera_center"].numpy(), -R.T @ t, atol=1e-6)
    np.testing.assert_allclose(cam["c2w"].numpy() @ w2c, np.eye(4), atol=1e-6)
    ours = syn.camera_from_w2c(w2c.astype(np.float64), cam["fovx"], cam["fovy"], H, W)   # the test/bench camera builder
    np.testing.assert_allclose(cam["full_proj_transform"].numpy(), ours["full_proj_transform"], rtol=1e-5, atol=1e-6)
    # a point on the optical axis projects to the image centre in NDC
    pw = np.linalg.inv(w2c) @ np.array([0, 0, 2.0, 1.0])
    clip = pw @ cam["full_proj_transform"].numpy()
    assert abs(clip[0] / clip[3]) < 1e-5 and abs(clip[1] / clip[3]) < 1e-5


@pytest.mark.gpu
def test_a_scene_read_from_ply_renders_like_the_tensors_it_was_written_from(device, tmp_path):
    from hugs_amd.renderer.gs_renderer import render
    cam0 = syn.pinhole_camera(96, 128)
    g = syn.scene_gaussians(3000, cam0, seed=3, sigma_px=3.0)
    shs = torch.from_numpy(g["shs"])
    p = {"xyz": torch.from_numpy(g["means3D"]), "features_dc": shs[:, :1].contiguous(), "features_rest": shs[:, 1:].contiguous(),
         "opacity": torch.logit(torch.from_numpy(g["opacities"]).clamp(1e-4, 1 - 1e-4)),
         "scaling": torch.log(torch.from_numpy(g["scales"])), "rotation": torch.from_numpy(g["rotations"])}
    path = str(tmp_path / "scene.ply")
    gio.write_gaussian_ply(path, **p)
    back = gio.read_gaussian_ply(path, max_sh_degree=3, device=device)
    data = {k: (torch.from_numpy(np.ascontiguousarray(v)).float().to(device) if isinstance(v, np.ndarray) else v) for k, v in cam0.items()}
    data["image_height"], data["image_width"] = 96, 128
    imgs = []
    for src in (gio.activated({**{k: v.to(device) for k, v in p.items()}, "active_sh_degree": 3}), gio.activated(back)):
        pkg = render(means3D=src["xyz"], feats=src["shs"], opacity=src["opacity"], scales=src["scales"], rotations=src["rotq"],
                     data=data, bg_color=torch.ones(3, device=device), active_sh_degree=3)
        imgs.append(pkg["render"])
    assert torch.equal(imgs[0], imgs[1]) and float(imgs[0].detach().std()) > 0.01
