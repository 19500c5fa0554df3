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


def pixel_grad(H, W, seed=1):
    """dL/dcolor ~ N(0,1) / (3 H W)."""
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((3, H, W)) / (3.0 * H * W)).astype(np.float32)
