#!/usr/bin/env python3
"""Generates tests/golden/reference_substeps.npz by IMPORTING the reference (read-only, from /root/reference)
in this container and recording inputs + outputs of the few functions that restate sub-steps of the
rasterizer hot path.  The reference itself never travels: only these vectors do.

    python tests/golden/make_golden.py            # writes tests/golden/reference_substeps.npz

What is recorded (all fp32, seeded):
  * eval_sh                     hugs/utils/spherical_harmonics.py:61-125   (degrees 0..3)
  * build_scaling_rotation + strip_symmetric -> Sigma3D 6-vector   hugs/utils/general.py:161-210
  * get_projection_matrix       hugs/utils/graphics.py:76-96
  * get_rotating_camera / get_static_camera dicts   hugs/datasets/utils.py:15-53,64-124
  * psnr                        hugs/utils/image.py:27-29
  * build_covariance_from_scaling_rotation(scaling, scaling_modifier, rotation)   hugs/models/scene.py:36-41
    (the one place the reference states how scale_modifier enters Sigma3D; reached through get_covariance, :144)
  * RGB2SH / SH2RGB             hugs/utils/spherical_harmonics.py:128-133   (the +0.5 / C0 colour convention)
  * the boundary transcript of render_human_scene (kwargs / settings / returned keys, shapes, dtypes)
    captured with a recording stand-in for diff_gaussian_rasterization   hugs/renderer/gs_renderer.py:20-161

The reference hard-codes device="cuda"; there is no GPU here, so torch factory functions are wrapped to
drop the device argument and Tensor.cuda() becomes the identity.  Missing third-party modules are
replaced by inert stubs that exist only inside this process.  Nothing is written to /root/reference.
"""
import importlib.abc
import importlib.machinery
import json
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_substeps.npz")

STUB_ROOTS = {"cv2", "pytorch3d", "loguru", "omegaconf", "torchvision", "trimesh", "smplx", "lpips", "open3d", "igl",
              "plyfile", "simple_knn", "joblib", "tqdm", "matplotlib", "imageio", "scipy_stub"}


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = types.ModuleType(spec.name)
        m.__path__ = []
        m.__getattr__ = lambda attr: mock.MagicMock(name=f"{spec.name}.{attr}")
        return m

    def exec_module(self, module):
        pass


def _shim_torch():
    torch.Tensor.cuda = lambda self, *a, **k: self
    for fname in ("zeros", "ones", "eye", "tensor", "empty", "zeros_like", "ones_like", "rand", "randn", "linspace",
                  "arange", "full"):
        orig = getattr(torch, fname)

        def wrapped(*a, _orig=orig, **k):
            if "device" in k:
                k.pop("device")
            return _orig(*a, **k)

        setattr(torch, fname, wrapped)
    orig_to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple(x for x in a if not (isinstance(x, str) and x.startswith("cuda")))
        if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
            k.pop("device")
        return orig_to(self, *a, **k) if (a or k) else self

    torch.Tensor.to = to


class _Recorder:
    """Stand-in for the diff_gaussian_rasterization module that records what crosses the boundary."""

    def __init__(self):
        self.calls = []
        mod = types.ModuleType("diff_gaussian_rasterization")
        rec = self

        from typing import NamedTuple

        class GaussianRasterizationSettings(NamedTuple):
            image_height: int
            image_width: int
            tanfovx: float
            tanfovy: float
            bg: torch.Tensor
            scale_modifier: float
            viewmatrix: torch.Tensor
            projmatrix: torch.Tensor
            sh_degree: int
            campos: torch.Tensor
            prefiltered: bool
            debug: bool

        class GaussianRasterizer(torch.nn.Module):
            def __init__(self, raster_settings):
                super().__init__()
                self.raster_settings = raster_settings

            def forward(self, **kw):
                s = self.raster_settings
                rec.calls.append({"settings": s, "kwargs": kw})
                P = kw["means3D"].shape[0]
                img = torch.zeros(3, s.image_height, s.image_width) + 0.0 * kw["means3D"].sum() + 0.0 * kw["means2D"].sum()
                return img, torch.ones(P, dtype=torch.int32)

        mod.GaussianRasterizationSettings = GaussianRasterizationSettings
        mod.GaussianRasterizer = GaussianRasterizer
        sys.modules["diff_gaussian_rasterization"] = mod


def _desc(v):
    if isinstance(v, torch.Tensor):
        return {"type": "tensor", "shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", ""),
                "requires_grad": bool(v.requires_grad)}
    if v is None:
        return {"type": "none"}
    return {"type": type(v).__name__, "value": v if isinstance(v, (int, float, bool, str)) else str(v)}


def main():
    assert os.path.isdir(REF), "this script runs only where /root/reference is mounted"
    sys.meta_path.insert(0, _StubFinder())
    _shim_torch()
    recorder = _Recorder()
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    out = {}

    # ---- SH evaluation (reference layout is [..., C, coeffs]; ours is [P, coeffs, C]) ----
    from hugs.utils import spherical_harmonics as rsh
    P = 257
    sh = rng.standard_normal((P, 16, 3)).astype(np.float32)
    dirs = rng.standard_normal((P, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    out["sh_coeffs"], out["sh_dirs"] = sh, dirs
    for deg in range(4):
        res = rsh.eval_sh(deg, torch.from_numpy(sh).permute(0, 2, 1), torch.from_numpy(dirs), rsh.C0, rsh.C1, rsh.C2,
                          rsh.C3, rsh.C4)
        out[f"sh_eval_deg{deg}"] = res.numpy()
    out["sh_C0"] = np.float32(rsh.C0.item())

    # ---- Sigma3D = (R S)(R S)^T packed (xx,xy,xz,yy,yz,zz); NB build_rotation NORMALISES q ----
    from hugs.utils import general as rgen
    s = np.exp(rng.normal(-2, 0.7, (P, 3))).astype(np.float32)
    q = rng.standard_normal((P, 4)).astype(np.float32)
    L = rgen.build_scaling_rotation(torch.from_numpy(s), torch.from_numpy(q))
    cov = rgen.strip_symmetric(L @ L.transpose(1, 2))
    out["cov_scales"], out["cov_quats_raw"], out["cov_packed"] = s, q, cov.numpy()

    # ---- Sigma3D with a scale modifier: SceneGS.setup_functions' nested build_covariance_from_scaling_rotation ----
    # scene.py:36-41, compiled from the source file (the module imports simple_knn / plyfile ...) with the reference's
    # own build_scaling_rotation / strip_symmetric (general.py) in its namespace; executed through the attribute the
    # reference itself uses (self.covariance_activation, scene.py:45,144)
    import ast as _ast

    def _method_src(path, cls, name, ns):
        tree = _ast.parse(open(path).read())
        node = next(f for c in tree.body if isinstance(c, _ast.ClassDef) and c.name == cls
                    for f in c.body if isinstance(f, _ast.FunctionDef) and f.name == name)
        node.decorator_list = []
        exec(compile(_ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
        return ns[name]

    setup_functions = _method_src(os.path.join(REF, "hugs/models/scene.py"), "SceneGS", "setup_functions",
                                  {"torch": torch, "build_scaling_rotation": rgen.build_scaling_rotation,
                                   "strip_symmetric": rgen.strip_symmetric, "inverse_sigmoid": rgen.inverse_sigmoid})
    holder = types.SimpleNamespace()
    setup_functions(holder)
    mods = np.array([0.5, 1.0, 1.7], np.float32)
    out["covmod_modifiers"] = mods
    out["covmod_packed"] = np.stack([holder.covariance_activation(torch.from_numpy(s), float(m), torch.from_numpy(q)).numpy()
                                     for m in mods])

    # ---- RGB <-> SH DC convention ----
    rgb = np.random.default_rng(123).uniform(-0.2, 1.2, (P, 3)).astype(np.float32)   # own generator: older vectors stay as they were
    out["rgb2sh_in"] = rgb
    out["rgb2sh_out"] = rsh.RGB2SH(torch.from_numpy(rgb)).numpy()
    out["sh2rgb_out"] = rsh.SH2RGB(torch.from_numpy(sh[:, 0])).numpy()      # of the DC coefficients recorded above

    # ---- projection matrix + camera dicts ----
    from hugs.utils import graphics as rgfx
    fovs = np.array([[0.4, 0.4], [1.0, 0.6], [1.2, 0.9]], np.float64)
    out["proj_fovs"] = fovs
    out["proj_mats"] = np.stack([rgfx.get_projection_matrix(0.01, 100.0, fx, fy).numpy() for fx, fy in fovs])
    from hugs.datasets import utils as rdu
    cams = rdu.get_rotating_camera(dist=5.0, img_size=512, nframes=4, device="cpu")
    for i, c in enumerate(cams):
        for k in ("world_view_transform", "full_proj_transform", "camera_center"):
            out[f"rotcam{i}_{k}"] = c[k].numpy().astype(np.float32)
    out["rotcam_fov"] = np.float64(cams[0]["fovx"])
    st = rdu.get_static_camera(img_size=256, fov=0.6, device="cpu")
    for k in ("world_view_transform", "full_proj_transform", "camera_center"):
        out[f"staticcam_{k}"] = st[k].numpy().astype(np.float32)

    # ---- PSNR ----
    from hugs.utils import image as rimg
    a = rng.uniform(0, 1, (2, 3, 32, 40)).astype(np.float32)
    b = np.clip(a + 0.05 * rng.standard_normal(a.shape), 0, 1).astype(np.float32)
    out["psnr_a"], out["psnr_b"] = a, b
    out["psnr_val"] = rimg.psnr(torch.from_numpy(a), torch.from_numpy(b)).numpy()

    # ---- boundary transcript ----
    from hugs.renderer import gs_renderer as rr

    def model(n, seed):
        g = torch.Generator().manual_seed(seed)
        return {"xyz": torch.randn(n, 3, generator=g).requires_grad_(True),
                "shs": torch.randn(n, 16, 3, generator=g).requires_grad_(True),
                "opacity": torch.rand(n, 1, generator=g).requires_grad_(True),
                "scales": torch.rand(n, 3, generator=g).requires_grad_(True),
                "rotq": torch.randn(n, 4, generator=g).requires_grad_(True), "active_sh_degree": 0}

    pkg = rr.render_human_scene(cams[1], model(7, 1), model(5, 2), bg_color=torch.ones(3),
                                human_bg_color=torch.zeros(3), render_mode="human_scene", render_human_separate=True)
    (pkg["render"].sum() + pkg["human_img"].sum()).backward()
    transcript = {"num_calls": len(recorder.calls), "calls": [], "returned": {k: _desc(v) for k, v in pkg.items()},
                  "viewspace_points_has_grad": pkg["viewspace_points"].grad is not None}
    for c in recorder.calls:
        transcript["calls"].append({"settings": {k: _desc(v) for k, v in c["settings"]._asdict().items()},
                                    "kwargs": {k: _desc(v) for k, v in c["kwargs"].items()}})
    out["boundary_transcript_json"] = np.frombuffer(json.dumps(transcript, sort_keys=True).encode(), dtype=np.uint8)

    # ---- densification statistics: the reference's own statements, executed ----
    # gs_trainer.py:406-411 (GaussianTrainer.scene_densification) and scene.py:460-462 (SceneGS.add_densification_stats)
    # (the two modules pull in half of the reference's dependency tree, so the two methods are compiled straight
    # from their source files -- the reference's own statements, untouched -- instead of importing the modules)
    import ast

    def _method_from_source(path, cls, name, ns):
        tree = ast.parse(open(path).read())
        node = next(f for c in tree.body if isinstance(c, ast.ClassDef) and c.name == cls
                    for f in c.body if isinstance(f, ast.FunctionDef) and f.name == name)
        node.decorator_list = []
        exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
        return ns[name]

    ns = {"torch": torch, "logger": mock.MagicMock()}
    SceneGS = types.SimpleNamespace(add_densification_stats=_method_from_source(
        os.path.join(REF, "hugs/models/scene.py"), "SceneGS", "add_densification_stats", dict(ns)))
    GaussianTrainer = types.SimpleNamespace(scene_densification=_method_from_source(
        os.path.join(REF, "hugs/trainer/gs_trainer.py"), "GaussianTrainer", "scene_densification", dict(ns)))
    n, extra = 300, 40          # the gradient tensor has `extra` more rows than the model (joint human+scene mode)
    g = torch.Generator().manual_seed(7)
    vpt = torch.zeros(n + extra, 3, requires_grad=True)
    vpt.grad = torch.randn(n + extra, 3, generator=g)
    radii = torch.randint(0, 60, (n,), generator=g, dtype=torch.int32)
    vis = radii > 12
    state = types.SimpleNamespace(max_radii2D=torch.rand(n, generator=g) * 50, xyz_gradient_accum=torch.rand(n, 1, generator=g),
                                  denom=torch.randint(0, 5, (n, 1), generator=g).float())
    out["dens_grad"], out["dens_radii"], out["dens_vis"] = vpt.grad.numpy().copy(), radii.numpy().copy(), vis.numpy().copy()
    out["dens_in_max_radii2D"], out["dens_in_accum"], out["dens_in_denom"] = (state.max_radii2D.numpy().copy(),
                                                                              state.xyz_gradient_accum.numpy().copy(),
                                                                              state.denom.numpy().copy())
    state.add_densification_stats = lambda v, f: SceneGS.add_densification_stats(state, v, f)
    trainer = types.SimpleNamespace(
        scene_gs=state, bg_color=torch.zeros(3),
        cfg=types.SimpleNamespace(scene=types.SimpleNamespace(densify_from_iter=10 ** 9, densification_interval=1,
                                                              opacity_reset_interval=10 ** 9)))
    with torch.no_grad():
        GaussianTrainer.scene_densification(trainer, visibility_filter=vis, radii=radii, viewspace_point_tensor=vpt, iteration=1)
    out["dens_out_max_radii2D"], out["dens_out_accum"], out["dens_out_denom"] = (state.max_radii2D.numpy().copy(),
                                                                                 state.xyz_gradient_accum.numpy().copy(),
                                                                                 state.denom.numpy().copy())

    # ---- SMPL neighbour-blended LBS quantities: the reference's own statements, executed ----
    # hugs/models/hugs_wo_trimlp.py:39-44 (batch_index_select), :47-85 (smpl_lbsmap_top_k), :88-119 (smpl_lbsweight_top_k),
    # compiled from the source file (the module imports pytorch3d, smplx, trimesh ...).  pytorch3d.ops.knn_points is a
    # pip dependency that is not in /root/reference: the functions are fed oracle/knn_oracle.py's restatement of its
    # contract, so these vectors pin every statement AFTER the search (and the search's inputs/outputs are recorded).
    sys.path.insert(0, os.path.join(os.path.dirname(OUT), "..", ".."))
    from oracle import knn_oracle

    def _function_from_source(path, name, ns):
        tree = ast.parse(open(path).read())
        node = next(f for f in tree.body if isinstance(f, ast.FunctionDef) and f.name == name)
        exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), ns)
        return ns[name]

    def _knn_points(p1, p2, K=1):
        d, i = knn_oracle.knn_points(p1[0].numpy(), p2[0].numpy(), K)
        return types.SimpleNamespace(dists=torch.from_numpy(d)[None], idx=torch.from_numpy(i)[None])

    kns = {"torch": torch, "knn_points": _knn_points}
    src = os.path.join(REF, "hugs/models/hugs_wo_trimlp.py")
    _function_from_source(src, "batch_index_select", kns)
    ref_lbsmap = _function_from_source(src, "smpl_lbsmap_top_k", kns)
    ref_lbsweight = _function_from_source(src, "smpl_lbsweight_top_k", kns)
    kr = np.random.default_rng(11)
    m, npts, J = 400, 250, 24
    templ = (kr.standard_normal((m, 3)) * np.array([0.25, 0.6, 0.15])).astype(np.float32)       # a body-shaped blob
    joints = templ[kr.choice(m, J, replace=False)]
    logits = -np.linalg.norm(templ[:, None, :] - joints[None], axis=-1) / 0.01                   # SMPL-like: near one-hot inside a part
    lbsw = np.exp(logits - logits.max(1, keepdims=True))
    lbsw = (lbsw / lbsw.sum(1, keepdims=True)).astype(np.float32)
    pts = (templ[kr.integers(0, m, npts)] + 0.02 * kr.standard_normal((npts, 3))).astype(np.float32)
    pts[:5] = templ[:5]                                                                          # exact hits: distance 0
    vT = np.tile(np.eye(4, dtype=np.float32), (m, 1, 1))
    vT[:, :3, :] += 0.1 * kr.standard_normal((m, 3, 4)).astype(np.float32)
    info = kr.standard_normal((m, 3)).astype(np.float32)
    kd, ki = knn_oracle.knn_points(pts, templ, 6)
    with torch.no_grad():
        d1, w1 = ref_lbsweight(torch.from_numpy(lbsw), torch.from_numpy(pts)[None], torch.from_numpy(templ)[None])
        d2, T2, i2 = ref_lbsmap(torch.from_numpy(lbsw), torch.from_numpy(vT)[None], torch.from_numpy(pts)[None],
                                torch.from_numpy(templ)[None], K=6, addition_info=torch.from_numpy(info)[None])
    out.update(knn_template=templ, knn_points=pts, knn_lbs_weights=lbsw, knn_verts_transform=vT, knn_addition_info=info,
               knn_search_dists=kd, knn_search_idx=ki,
               knn_lbsweight_dist=d1[0].numpy(), knn_lbsweight_weights=w1[0].numpy(),
               knn_lbsmap_dist=d2[0].numpy(), knn_lbsmap_transform=T2[0].numpy(), knn_lbsmap_info=i2[0].numpy())

    # ---- learned LBS skinning (second half): lbs_extra, hugs/models/modules/lbs.py:19-73, compiled
    # from its source file and executed (forward AND autograd backward), plus the rotation product of hugs_trimlp.py:517.
    # smplx (pip dependency) is absent: lbs_extra's only use of it on the release path (disable_posedirs: true) is a
    # batch_rodrigues call whose result is discarded; it is given oracle/lbs_oracle.py's restatement of the published
    # formula, which therefore only shapes the posedirs-enabled vectors below.
    from oracle import lbs_oracle
    lns = {"torch": torch, "Tensor": torch.Tensor,
           "batch_rodrigues": lambda r: torch.from_numpy(lbs_oracle.batch_rodrigues(r.detach().numpy().astype(np.float64), np.float64)).to(r.dtype)}
    ref_lbs_extra = _function_from_source(os.path.join(REF, "hugs/models/modules/lbs.py"), "lbs_extra", lns)
    lr = np.random.default_rng(17)
    nJ, nV = 24, 96
    A_np = np.tile(np.eye(4, dtype=np.float32), (nJ, 1, 1))
    A_np[:, :3, :] += 0.3 * lr.standard_normal((nJ, 3, 4)).astype(np.float32)                 # rigid-ish joint transforms
    logit = 4.0 * lr.standard_normal((nV, nJ))
    Wl = (np.exp(logit) / np.exp(logit).sum(1, keepdims=True)).astype(np.float32)             # softmax(x / 0.1)-like: peaky rows
    v_np = (lr.standard_normal((nV, 3)) * np.array([0.25, 0.6, 0.15])).astype(np.float32)
    R_np = lbs_oracle.batch_rodrigues(lr.standard_normal((nV, 3)).astype(np.float32))          # per-Gaussian rotation matrices
    pose_np = (0.4 * lr.standard_normal((1, nJ * 3))).astype(np.float32)
    posedirs_np = (0.01 * lr.standard_normal(((nJ - 1) * 9, nV * 3))).astype(np.float32)
    g_verts, g_T = lr.standard_normal((nV, 3)).astype(np.float32), lr.standard_normal((nV, 4, 4)).astype(np.float32)
    g_rot = lr.standard_normal((nV, 3, 3)).astype(np.float32)
    out.update(lbs_A=A_np, lbs_weights=Wl, lbs_v=v_np, lbs_rotmat=R_np, lbs_pose=pose_np, lbs_posedirs=posedirs_np,
               lbs_g_verts=g_verts, lbs_g_T=g_T, lbs_g_rot=g_rot)
    for tag, disable in (("", True), ("_posedirs", False)):
        tA, tW, tv = (torch.from_numpy(x.copy()).requires_grad_(True) for x in (A_np, Wl, v_np))
        tR = torch.from_numpy(R_np.copy()).requires_grad_(True)
        verts, _, T_, v_posed, _ = ref_lbs_extra(tA[None], tv[None], torch.from_numpy(posedirs_np), tW, torch.from_numpy(pose_np),
                                                 disable_posedirs=disable, pose2rot=True)
        rot = T_[0][:, :3, :3] @ tR                                                             # hugs_trimlp.py:517
        ((verts[0] * torch.from_numpy(g_verts)).sum() + (T_[0] * torch.from_numpy(g_T)).sum() + (rot * torch.from_numpy(g_rot)).sum()).backward()
        out.update({f"lbs{tag}_verts": verts[0].detach().numpy(), f"lbs{tag}_T": T_[0].detach().numpy(), f"lbs{tag}_rot": rot.detach().numpy(),
                    f"lbs{tag}_v_posed": v_posed[0].detach().numpy(), f"lbs{tag}_dA": tA.grad.numpy(), f"lbs{tag}_dW": tW.grad.numpy(),
                    f"lbs{tag}_dv": tv.grad.numpy(), f"lbs{tag}_dR": tR.grad.numpy()})

    # ---- PLY attribute order: SceneGS.construct_list_of_attributes, scene.py:229-241, executed ----
    # (save_ply / load_ply themselves need the `plyfile` package, which is not installed: only the attribute list,
    # which fixes the on-disk column order, can be produced by the reference here)
    cla = _method_from_source(os.path.join(REF, "hugs/models/scene.py"), "SceneGS", "construct_list_of_attributes", {})
    fake = types.SimpleNamespace(_features_dc=torch.zeros(2, 1, 3), _features_rest=torch.zeros(2, 15, 3),
                                 _scaling=torch.zeros(2, 3), _rotation=torch.zeros(2, 4))
    out["ply_attribute_names_json"] = np.frombuffer(json.dumps(cla(fake)).encode(), dtype=np.uint8)

    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
