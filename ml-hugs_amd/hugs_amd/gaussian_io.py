"""On-disk Gaussian format and camera construction: what it takes to put a trained HUGS / 3DGS
scene in front of the rasterizer instead of synthetic inputs.

* `write_gaussian_ply` / `read_gaussian_ply`: the PLY layout of `SceneGS.save_ply` / `load_ply`
  (/root/reference/hugs/models/scene.py:229-308): one `vertex` element, all `float` properties, in the order of
  `construct_list_of_attributes` (:229-241) -- x y z nx ny nz f_dc_* f_rest_* opacity scale_* rot_* -- with the
  reference's storage conventions: opacity as a logit, scales as logs, quaternions raw (w,x,y,z), SH as
  [P,1,3] + [P,15,3] tensors flattened channel-major (`transpose(1,2).flatten(1)`).  The reference goes through the
  `plyfile` package (absent here); this module writes/reads binary_little_endian PLY directly (the same bytes
  `PlyData([el]).write` produces for such an element) and also reads ASCII PLY.
* `activated`: what `SceneGS.forward` (:147-160) hands the renderer -- exp / normalize / sigmoid / cat(dc, rest).
* `camera_from_colmap`: the camera dict of `NeumanDataset.__getitem__` (/root/reference/hugs/datasets/neuman.py:346-375)
  from an intrinsic matrix and a world-to-camera pose.

Pure host code (numpy + torch); no GPU kernel.
"""
import math
import os

import numpy as np
import torch


SH_C0 = 0.28209479177387814


def RGB2SH(rgb):
    """/root/reference/hugs/utils/spherical_harmonics.py:128-129 -- the DC coefficient that renders as `rgb`
    (used when a scene is created from a coloured point cloud, scene.py:166-194)."""
    return (rgb - 0.5) / SH_C0


def SH2RGB(sh):
    """spherical_harmonics.py:132-133 -- what a degree-0 coefficient renders as (before the rasterizer's clamp at 0)."""
    return sh * SH_C0 + 0.5


def attribute_names(n_dc=3, n_rest=45, n_scale=3, n_rot=4):
    """scene.py:229-241 (construct_list_of_attributes)."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    return names


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def write_gaussian_ply(path, xyz, features_dc, features_rest, opacity, scaling, rotation):
    """scene.py:243-260.  features_dc [P,1,3], features_rest [P,K-1,3] (the module's parameter layout), opacity [P,1]
    (logit), scaling [P,3] (log), rotation [P,4]."""
    if os.path.dirname(path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
    xyz = _np(xyz).astype(np.float32)
    f_dc = torch.as_tensor(_np(features_dc)).transpose(1, 2).flatten(start_dim=1).contiguous().numpy().astype(np.float32)
    f_rest = torch.as_tensor(_np(features_rest)).transpose(1, 2).flatten(start_dim=1).contiguous().numpy().astype(np.float32)
    cols = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, _np(opacity).reshape(len(xyz), -1).astype(np.float32),
                           _np(scaling).astype(np.float32), _np(rotation).astype(np.float32)), axis=1)
    names = attribute_names(f_dc.shape[1], f_rest.shape[1], _np(scaling).shape[1], _np(rotation).shape[1])
    assert cols.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {len(xyz)}\n" + \
        "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(cols, dtype="<f4").tobytes())


_PLY_TYPES = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1", "char": "i1",
              "int8": "i1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4",
              "uint": "u4", "uint32": "u4"}


def read_ply_vertices(path):
    """The `vertex` element of a PLY file as a dict {property: 1-D numpy array} (ascii or binary, scalar properties)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex, first = None, None, [], False, True
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if first and tok[1] != "vertex":
                    raise ValueError(f"{path}: the vertex element must come first")
                in_vertex, first = tok[1] == "vertex", False
                if in_vertex:
                    count = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties in the vertex element are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if count is None:
            raise ValueError(f"{path}: no vertex element")
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=count, ndmin=2, dtype=np.float64) if count else np.zeros((0, len(props)))
            return {n: rows[:, k].astype(t) for k, (n, t) in enumerate(props)}
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, order + t) for n, t in props])
        data = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
        return {n: np.ascontiguousarray(data[n]) for n, _ in props}


def read_gaussian_ply(path, max_sh_degree=3, device="cpu"):
    """scene.py:267-306: -> dict of fp32 tensors in the module's parameter layout (xyz, features_dc [P,1,3],
    features_rest [P,K-1,3], opacity [P,1], scaling [P,3], rotation [P,4])."""
    v = read_ply_vertices(path)
    xyz = np.stack((v["x"], v["y"], v["z"]), axis=1)
    opacities = np.asarray(v["opacity"])[..., np.newaxis]
    features_dc = np.zeros((xyz.shape[0], 3, 1))
    for c in range(3):
        features_dc[:, c, 0] = v[f"f_dc_{c}"]
    by_index = lambda prefix: sorted((n for n in v if n.startswith(prefix)), key=lambda x: int(x.split("_")[-1]))
    extra = by_index("f_rest_")
    if len(extra) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError(f"{path}: {len(extra)} f_rest_* properties do not match SH degree {max_sh_degree}")
    features_extra = np.stack([v[n] for n in extra], axis=1) if extra else np.zeros((xyz.shape[0], 0))
    features_extra = features_extra.reshape((features_extra.shape[0], 3, (max_sh_degree + 1) ** 2 - 1))
    scales = np.stack([v[n] for n in by_index("scale_")], axis=1)
    rots = np.stack([v[n] for n in by_index("rot")], axis=1)
    t = lambda a: torch.tensor(a, dtype=torch.float, device=device)
    return {"xyz": t(xyz), "features_dc": t(features_dc).transpose(1, 2).contiguous(),
            "features_rest": t(features_extra).transpose(1, 2).contiguous(), "opacity": t(opacities),
            "scaling": t(scales), "rotation": t(rots), "active_sh_degree": max_sh_degree}


def activated(params):
    """scene.py:147-160 (SceneGS.forward): the dict the renderer consumes."""
    return {"xyz": params["xyz"], "scales": torch.exp(params["scaling"]),
            "rotq": torch.nn.functional.normalize(params["rotation"]),
            "shs": torch.cat((params["features_dc"], params["features_rest"]), dim=1),
            "opacity": torch.sigmoid(params["opacity"]), "active_sh_degree": params["active_sh_degree"]}


def projection_matrix(znear, zfar, fovX, fovY):
    """hugs/utils/graphics.py:76-96 (get_projection_matrix), checked against the reference in tests/golden."""
    t, r = math.tan(fovY / 2) * znear, math.tan(fovX / 2) * znear
    P = torch.zeros(4, 4)
    P[0, 0], P[1, 1] = 2.0 * znear / (2 * r), 2.0 * znear / (2 * t)
    P[3, 2] = 1.0
    P[2, 2], P[2, 3] = zfar / (zfar - znear), -(zfar * znear) / (zfar - znear)
    return P


def camera_from_colmap(intrinsic_matrix, world_to_camera, height, width, znear=0.01, zfar=100.0):
    """neuman.py:346-375: K [3,3], world_to_camera [4,4] (column-vector convention, as COLMAP / NeuMan store it)."""
    K = np.asarray(intrinsic_matrix, dtype=np.float64)
    fovx = 2 * np.arctan(width / (2 * K[0, 0]))
    fovy = 2 * np.arctan(height / (2 * K[1, 1]))
    w2c = np.asarray(world_to_camera)
    world_view_transform = torch.from_numpy(w2c).T
    c2w = torch.from_numpy(np.linalg.inv(w2c))
    proj = projection_matrix(znear=znear, zfar=zfar, fovX=fovx, fovY=fovy).transpose(0, 1).to(world_view_transform.dtype)
    full_proj_transform = (world_view_transform.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    camera_center = world_view_transform.inverse()[3, :3]
    return {"fovx": fovx, "fovy": fovy, "image_height": height, "image_width": width,
            "world_view_transform": world_view_transform, "c2w": c2w, "full_proj_transform": full_proj_transform,
            "camera_center": camera_center, "cam_intrinsics": torch.from_numpy(K).float(), "near": znear, "far": zfar}
