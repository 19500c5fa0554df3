"""Renderer adapter: the caller side of the drop-in boundary.

Mirrors the interface of /root/reference/hugs/renderer/gs_renderer.py (`render` :103-161,
`render_human_scene` :20-100): same function names, argument meaning, returned dict keys, dtypes and
the behaviours the trainer relies on:
  * default background is black, created on the device of the Gaussians        (:104-105)
  * `viewspace_points` is a non-leaf zero tensor that retains its grad, and is passed as means2D so
    the rasterizer's dL/d(screen-space xy) lands in `viewspace_points.grad`    (:107-113)
  * tanfov is computed on the host in double precision                           (:116-117)
  * `feats.ndim == 2` selects precomputed colours, otherwise SH                  (:119-123)
  * prefiltered=False, debug=False; image clamped to [0,1]; visibility = radii>0 (:137-138,153,159)
  * joint mode concatenates human first, scene second, takes the active SH degree from the human
    model, optionally renders the human alone on its own background, and slices radii/visibility
    per model                                                                    (:32-52,68-98)
"""
import math
import os

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

_FIELDS = (("shs", "feats"), ("xyz", "means3D"), ("opacity", "opacity"), ("scales", "scales"), ("rotq", "rotations"))


def _gather(human_gs_out, scene_gs_out, render_mode):
    if render_mode == "human_scene":
        out = {dst: torch.cat([human_gs_out[src], scene_gs_out[src]], dim=0) for src, dst in _FIELDS}
        out["active_sh_degree"] = human_gs_out["active_sh_degree"]
    elif render_mode == "human":
        out = {dst: human_gs_out[src] for src, dst in _FIELDS}
        out["active_sh_degree"] = human_gs_out["active_sh_degree"]
    elif render_mode == "scene":
        out = {dst: scene_gs_out[src] for src, dst in _FIELDS}
        out["active_sh_degree"] = scene_gs_out["active_sh_degree"]
    else:
        raise ValueError(f"Unknown render mode: {render_mode}")
    return out


# The joint render and the separate human-only render of one training step (gs_renderer.py:56 and :69) are independent:
# the second one goes to a side HIP stream, so that its latency-bound binning overlaps the first one's blending (forward
# and, through autograd's stream bookkeeping, backward too).  Same results; HGS_CONCURRENT_RENDERS=0 turns it off.
_CONCURRENT_RENDERS = os.environ.get("HGS_CONCURRENT_RENDERS", "1") != "0"
_side_streams = {}


def _side_stream(device):
    if device not in _side_streams:
        _side_streams[device] = torch.cuda.Stream(device)
    return _side_streams[device]


def render_human_scene(data, human_gs_out, scene_gs_out, bg_color, human_bg_color=None, scaling_modifier=1.0,
                       render_mode="human_scene", render_human_separate=False):
    g = _gather(human_gs_out, scene_gs_out, render_mode)
    separate = render_human_separate and render_mode == "human_scene"
    human_pkg = None

    def human_only():
        h = _gather(human_gs_out, None, "human")
        return render(means3D=h["means3D"], feats=h["feats"], opacity=h["opacity"], scales=h["scales"],
                      rotations=h["rotations"], data=data, scaling_modifier=scaling_modifier,
                      bg_color=human_bg_color if human_bg_color is not None else bg_color,
                      active_sh_degree=h["active_sh_degree"])

    device = g["means3D"].device
    side = main = None
    if separate and _CONCURRENT_RENDERS and device.type == "cuda":
        main, side = torch.cuda.current_stream(device), _side_stream(device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            human_pkg = human_only()
            # main-stream tensors read (and saved for backward) by side-stream kernels: tell the caching allocator, so
            # that a caller dropping them early cannot have their blocks recycled under those kernels
            used = [human_gs_out[src] for src, _ in _FIELDS] + [bg_color, human_bg_color] + \
                   [data[k] for k in ("world_view_transform", "full_proj_transform", "camera_center")]
            for t in used:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(side)
    pkg = render(means3D=g["means3D"], feats=g["feats"], opacity=g["opacity"], scales=g["scales"],
                 rotations=g["rotations"], data=data, scaling_modifier=scaling_modifier, bg_color=bg_color,
                 active_sh_degree=g["active_sh_degree"])

    if separate:
        if human_pkg is None:
            human_pkg = human_only()
        else:
            main.wait_stream(side)
            for t in (human_pkg["render"], human_pkg["radii"], human_pkg["visibility_filter"]):
                t.record_stream(main)  # allocated on the side stream, consumed on the caller's
        pkg["human_img"] = human_pkg["render"]
        pkg["human_visibility_filter"] = human_pkg["visibility_filter"]
        pkg["human_radii"] = human_pkg["radii"]

    if render_mode == "human":
        pkg["human_visibility_filter"] = pkg["visibility_filter"]
        pkg["human_radii"] = pkg["radii"]
    elif render_mode == "human_scene":
        n_h, n_s = human_gs_out["xyz"].shape[0], scene_gs_out["xyz"].shape[0]
        pkg["scene_visibility_filter"] = pkg["visibility_filter"][n_h:]
        pkg["scene_radii"] = pkg["radii"][n_h:]
        if "human_visibility_filter" not in pkg:
            pkg["human_visibility_filter"] = pkg["visibility_filter"][:-n_s]
            pkg["human_radii"] = pkg["radii"][:-n_s]
    else:  # scene
        pkg["scene_visibility_filter"] = pkg["visibility_filter"]
        pkg["scene_radii"] = pkg["radii"]
    return pkg


def render(means3D, feats, opacity, scales, rotations, data, scaling_modifier=1.0, bg_color=None,
           active_sh_degree=0):
    device = means3D.device
    if bg_color is None:
        bg_color = torch.zeros(3, dtype=torch.float32, device=device)

    # gradient sink for dL/d(screen-space mean): non-leaf, so it must retain its grad explicitly
    screenspace_points = torch.zeros_like(means3D, dtype=means3D.dtype, requires_grad=True, device=device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass

    settings = GaussianRasterizationSettings(
        image_height=int(data["image_height"]),
        image_width=int(data["image_width"]),
        tanfovx=math.tan(data["fovx"] * 0.5),
        tanfovy=math.tan(data["fovy"] * 0.5),
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=data["world_view_transform"],
        projmatrix=data["full_proj_transform"],
        sh_degree=active_sh_degree,
        campos=data["camera_center"],
        prefiltered=False,
        debug=False,
    )
    is_rgb = feats.dim() == 2
    image, radii = GaussianRasterizer(raster_settings=settings)(
        means3D=means3D,
        means2D=screenspace_points,
        shs=None if is_rgb else feats,
        colors_precomp=feats if is_rgb else None,
        opacities=opacity,
        scales=scales,
        rotations=rotations,
        clamp_output=True,   # the reference's torch.clamp(rendered_image, 0.0, 1.0) (:153), fused into the blend kernels
    )
    return {
        "render": image,
        "viewspace_points": screenspace_points,
        "visibility_filter": radii > 0,
        "radii": radii,
    }
