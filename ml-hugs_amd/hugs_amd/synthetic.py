"""Seeded synthetic cameras and Gaussian sets for tests and bench.py.

Camera dicts carry exactly the keys the reference's datasets put into `data`
(/root/reference/hugs/datasets/neuman.py:346-362, /root/reference/hugs/datasets/utils.py:15-53):
fovx, fovy, image_height, image_width, world_view_transform (= W2C^T), full_proj_transform
(= W2C^T @ P^T), camera_center (= inv(world_view)[3,:3]).  All numpy, fp32; the caller moves
them to the device.
"""
import math

import numpy as np


def projection_matrix(znear, zfar, fovx, fovy):
    """Same matrix as /root/reference/hugs/utils/graphics.py:76-96 (z_sign = +1, P[3,2] = 1)."""
    tan_y, tan_x = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = tan_y * znear, tan_x * znear
    bottom, left = -top, -right
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def camera_from_w2c(w2c, fovx, fovy, height, width, znear=0.01, zfar=100.0):
    """Build the reference's camera dict from a 4x4 world-to-camera matrix (column-vector form)."""
    w2c = np.asarray(w2c, np.float32)
    world_view = np.ascontiguousarray(w2c.T)
    proj = projection_matrix(znear, zfar, fovx, fovy).T
    full = (world_view @ proj).astype(np.float32)
    center = np.linalg.inv(world_view.astype(np.float64))[3, :3].astype(np.float32)
    return {
        "fovx": float(fovx), "fovy": float(fovy), "image_height": int(height), "image_width": int(width),
        "world_view_transform": world_view, "full_proj_transform": full, "camera_center": center,
        "near": znear, "far": zfar,
    }


def pinhole_camera(height, width, focal_frac=0.9, w2c=None):
    """Pinhole with f = focal_frac * W (fov_x ~ 58 deg at 0.9); identity W2C unless given."""
    f = focal_frac * width
    fovx = 2.0 * math.atan(width / (2.0 * f))
    fovy = 2.0 * math.atan(height / (2.0 * f))
    return camera_from_w2c(np.eye(4) if w2c is None else w2c, fovx, fovy, height, width)


def rotating_camera(i, nframes, dist=5.0, fov=0.4, img_size=512):
    """Frame i of an orbit about the y axis at distance `dist`, looking at the origin
    (the canonical-view rig of /root/reference/hugs/datasets/utils.py:64-124)."""
    az = 2.0 * math.pi * i / max(nframes - 1, 1)
    c, s = math.cos(-az), math.sin(-az)
    Ry = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)
    cam_pos = Ry @ np.array([0.0, 0.0, dist])
    fwd = -cam_pos / np.linalg.norm(cam_pos)
    up = np.array([0.0, -1.0, 0.0])
    right = np.cross(up, fwd)
    right /= np.linalg.norm(right)
    up2 = np.cross(fwd, right)
    R = np.stack([right, up2, fwd], 0)  # rows: camera axes in world
    w2c = np.eye(4)
    w2c[:3, :3] = R
    w2c[:3, 3] = -R @ cam_pos
    return camera_from_w2c(w2c, fov, fov, img_size, img_size)


def scene_gaussians(P, cam, seed=0, sh_coeffs=16, sigma_px=4.5, nonunit_quat=False, ref_P=200_000, cluster=0.0):
    """Random Gaussians filling the view frustum of `cam` (camera must be the identity pose).

    Screen-space size is controlled: the world scale of each Gaussian is proportional to its
    depth so that its projected sigma is ~ sigma_px * lognormal(0.5) pixels, divided by
    sqrt(P / ref_P) so total coverage stays comparable across the P sweep.  `cluster` = fraction of the Gaussians
    packed into a person-sized blob in the middle of the view (a human in front of a scene) instead of spread uniformly.
    """
    rng = np.random.default_rng(seed)
    H, W = cam["image_height"], cam["image_width"]
    tanx, tany = math.tan(cam["fovx"] / 2), math.tan(cam["fovy"] / 2)
    z = rng.uniform(1.0, 20.0, P)
    x = rng.uniform(-1.1, 1.1, P) * z * tanx
    y = rng.uniform(-1.1, 1.1, P) * z * tany
    if cluster > 0.0:
        nc = int(P * cluster)
        z[:nc] = rng.uniform(4.0, 5.0, nc)
        x[:nc] = 0.18 * rng.standard_normal(nc) * z[:nc] * tanx
        y[:nc] = 0.45 * rng.standard_normal(nc) * z[:nc] * tany
    means = np.stack([x, y, z], 1).astype(np.float32)
    f = W / (2.0 * tanx)
    base = z[:, None] * (sigma_px / f) / math.sqrt(max(P, 1) / ref_P)
    scales = (base * np.exp(0.5 * rng.standard_normal((P, 3)))).astype(np.float32)
    q = rng.standard_normal((P, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    if nonunit_quat:
        q *= rng.uniform(0.8, 1.2, (P, 1))
    opac = (1.0 / (1.0 + np.exp(-1.5 * rng.standard_normal((P, 1))))).astype(np.float32)
    shs = np.zeros((P, sh_coeffs, 3), np.float32)
    shs[:, 0] = rng.standard_normal((P, 3))
    if sh_coeffs > 1:
        shs[:, 1:] = 0.1 * rng.standard_normal((P, sh_coeffs - 1, 3))
    return {"means3D": means, "scales": scales, "rotations": q.astype(np.float32), "opacities": opac, "shs": shs}


def trained_scene_gaussians(P, cam, seed=0, sh_coeffs=16, human=110_210, median_px=1.6, ref_P=200_000):
    """A scene shaped like what HUGS actually renders after some thousand steps, not like `scene_gaussians`' uniform
    fog: the reference builds its scene with create_from_pcd from a COLMAP cloud -- points on SURFACES -- and then clones,
    splits and prunes (/root/reference/hugs/models/scene.py:166-194,441-458), and resets opacities to <= 0.01 every
    `opacity_reset_interval` steps (cfg_files/release/neuman/hugs_scene.yaml:112).  Here:
      * depths clustered on a few surfaces -- a ground plane and five walls / objects, a few centimetres thick;
      * heavy-tailed sizes: projected sigma ~ median_px * lognormal(1.2), so 1-2 % of the splats are 100 px and more across
        (the big background splats every trained 3DGS scene keeps) while most are a pixel or two;
      * 30 % of the opacities <= 0.02 (the population just after a reset), the rest sigmoid(N(0, 1.5));
      * `human` Gaussians on a person-sized body SHELL 4-5 units in front of the camera (a thin surface, not a filled blob:
        what the SMPL-initialised human model is), in front of it all.
    Same dictionary as scene_gaussians; the human comes FIRST (the order render_human_scene concatenates in)."""
    rng = np.random.default_rng(seed)
    H, W = cam["image_height"], cam["image_width"]
    tanx, tany = math.tan(cam["fovx"] / 2), math.tan(cam["fovy"] / 2)
    f = W / (2.0 * tanx)
    # ---- scene: surfaces
    kind = rng.integers(0, 6, P)
    u, v = rng.uniform(-1.1, 1.1, P), rng.uniform(-1.1, 1.1, P)
    z = np.empty(P)
    ground = kind == 0
    z[ground] = rng.uniform(2.0, 20.0, ground.sum())                       # the ground: all depths, below the horizon
    v[ground] = 0.25 + 0.85 * (2.0 / z[ground])                            # (a plane 0.5 units under the camera, roughly)
    wall_z = np.array([6.0, 9.0, 12.0, 15.0, 19.0])
    for k in range(1, 6):
        m = kind == k
        z[m] = wall_z[k - 1] + 0.03 * rng.standard_normal(m.sum())        # a wall: a few centimetres thick
        u[m] = np.clip((k - 3) * 0.35 + 0.45 * rng.standard_normal(m.sum()), -1.1, 1.1)
    x, y = u * z * tanx, v * z * tany
    sig_px = median_px * np.exp(1.2 * rng.standard_normal(P)) / math.sqrt(max(P, 1) / ref_P)
    base = z * sig_px / f
    scales = base[:, None] * np.exp(0.35 * rng.standard_normal((P, 3)))
    opac = 1.0 / (1.0 + np.exp(-1.5 * rng.standard_normal(P)))
    reset = rng.uniform(size=P) < 0.30
    opac[reset] = rng.uniform(0.004, 0.02, reset.sum())
    # ---- human: a body shell (ellipsoid surface) ~1.7 units tall at depth 4.5
    Ph = int(human)
    th, ph = rng.uniform(0, 2 * math.pi, Ph), np.arccos(rng.uniform(-1, 1, Ph))
    hx = 0.28 * np.sin(ph) * np.cos(th)
    hy = 0.85 * np.cos(ph) + 0.05
    hz = 4.5 + 0.18 * np.sin(ph) * np.sin(th) + 0.004 * rng.standard_normal(Ph)
    h_sig = 0.0045 * np.exp(0.3 * rng.standard_normal((Ph, 3))) / math.sqrt(max(Ph, 1) / 110_210)
    h_op = rng.uniform(0.3, 0.98, Ph)
    n = P + Ph
    means = np.concatenate([np.stack([hx, hy, hz], 1), np.stack([x, y, z], 1)], 0).astype(np.float32)
    sc = np.concatenate([h_sig, scales], 0).astype(np.float32)
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q *= rng.uniform(0.9, 1.1, (n, 1))                                      # (HUGS does not normalise the human's quaternions)
    shs = np.zeros((n, sh_coeffs, 3), np.float32)
    shs[:, 0] = rng.standard_normal((n, 3))
    if sh_coeffs > 1:
        shs[:, 1:] = 0.1 * rng.standard_normal((n, sh_coeffs - 1, 3))
    return {"means3D": means, "scales": sc, "rotations": q.astype(np.float32),
            "opacities": np.concatenate([h_op, opac], 0).astype(np.float32)[:, None], "shs": shs, "n_human": Ph}


def pixel_grad(H, W, seed=1):
    """dL/dcolor ~ N(0,1) / (3 H W)."""
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((3, H, W)) / (3.0 * H * W)).astype(np.float32)
