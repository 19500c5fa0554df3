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
  * joint mode renders human first, scene second, takes the active SH degree from the human
    model, optionally renders the human alone on its own background, and slices radii/visibility
    per model                                                                    (:32-52,68-98)
The one structural difference: the reference concatenates the two models' five tensors per step (:33-37) and autograd
splits the gradients again; here the scene's tensors travel as the rasterizer's SECOND SEGMENT (hgs_segment), read and
differentiated in place -- Gaussian indices, radii, viewspace_points and every value are those of the concatenated call
(HGS_JOINT_CONCAT=1 keeps the reference's torch.cat for A/B runs).
"""
import math
import os

import torch

import diff_gaussian_rasterization as _dgr
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, rasterize_deferred

_FIELDS = (("shs", "feats"), ("xyz", "means3D"), ("opacity", "opacity"), ("scales", "scales"), ("rotq", "rotations"))


_JOINT_CONCAT = os.environ.get("HGS_JOINT_CONCAT", "0") == "1"
_VIEWSPACE_NONLEAF = os.environ.get("HGS_VIEWSPACE_NONLEAF", "0") == "1"
_FUSED_VISIBILITY = os.environ.get("HGS_FUSED_VISIBILITY", "1") != "0"


# Round 5: one C++ call per frame.  render() -- viewspace tensor, settings, rasterization, visibility -- and the two renders of a
# training step (render_human_scene with render_human_separate) each are ONE call into the C++ binding (csrc_torch/hgs_torch.cpp:
# render / render_pair) with one hand-made autograd node behind it: the frames HUGS renders most take ~120-150 us of kernels, and
# the Python statements below cost as much on a slow host.  Same kernels, same values.  HGS_FRAME_CALL=0 (or any of the A/B
# switches above that change what a frame consists of) keeps the statement-by-statement path.
_FRAME_CALL = os.environ.get("HGS_FRAME_CALL", "1") != "0"


def _frame_call(device):
    if not _FRAME_CALL or _VIEWSPACE_NONLEAF or not _FUSED_VISIBILITY or device.type != "cuda":
        return None
    return _dgr._load_cpp()


_MODEL_KEYS = ("xyz", "shs", "opacity", "scales", "rotq")   # (means3D, feats, opacity, scales, rotations) of a model's output dict


def _two_segments(human_gs_out, scene_gs_out):
    """Can the joint render pass the scene as the rasterizer's second segment?  (Both models non-empty and holding the same
    kind of features; otherwise -- and with HGS_JOINT_CONCAT=1 -- the tensors are concatenated as the reference does.)"""
    h, sc = human_gs_out, scene_gs_out
    return (not _JOINT_CONCAT and h["xyz"].shape[0] > 0 and sc["xyz"].shape[0] > 0 and h["shs"].dim() == sc["shs"].dim()
            and h["xyz"].device == sc["xyz"].device)


def _gather(human_gs_out, scene_gs_out, render_mode, concat=True):
    if render_mode == "human_scene" and not concat:
        out = {dst: human_gs_out[src] for src, dst in _FIELDS}
        out["second"] = {dst: scene_gs_out[src] for src, dst in _FIELDS}
        out["active_sh_degree"] = human_gs_out["active_sh_degree"]
    elif render_mode == "human_scene":
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
    g = _gather(human_gs_out, scene_gs_out, render_mode,
                concat=not (render_mode == "human_scene" and _two_segments(human_gs_out, scene_gs_out)))
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
    cpp = _frame_call(device) if (separate and _CONCURRENT_RENDERS and "second" in g) else None
    if cpp is not None:
        # both renders of the step in one call, one autograd node: the human-only frame runs on a library-side stream under the joint
        # one (forward and backward) and its gradients of the human tensors are summed inside the joint frame's per-Gaussian kernel
        out = cpp.render_pair([human_gs_out[k] for k in _MODEL_KEYS], [scene_gs_out[k] for k in _MODEL_KEYS], bg_color,
                              human_bg_color if human_bg_color is not None else bg_color, data["world_view_transform"],
                              data["full_proj_transform"], data["camera_center"], int(data["image_height"]), int(data["image_width"]),
                              float(data["fovx"]), float(data["fovy"]), float(scaling_modifier), int(g["active_sh_degree"]))
        n_h = human_gs_out["xyz"].shape[0]
        return {"render": out[0], "viewspace_points": out[3], "visibility_filter": out[2], "radii": out[1],
                "human_img": out[4], "human_visibility_filter": out[6], "human_radii": out[5],
                "scene_visibility_filter": out[2][n_h:], "scene_radii": out[1][n_h:]}
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
                 active_sh_degree=g["active_sh_degree"], second=g.get("second"))

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
           active_sh_degree=0, second=None):
    """`second` (not in the reference's signature): a dict {means3D, feats, opacity, scales, rotations} with a second model's
    Gaussians, rendered behind the first in index order -- what the reference gets by concatenating (:33-37)."""
    device = means3D.device
    if bg_color is None:
        bg_color = torch.zeros(3, dtype=torch.float32, device=device)

    # gradient sink for dL/d(screen-space mean), one row per Gaussian (of both models in a joint render).  The reference builds
    # it as `zeros_like(means3D, requires_grad=True) + 0` plus retain_grad() (:107-113): a NON-LEAF zero tensor -- two
    # elementwise kernels per render, and the retain_grad hook clones the gradient in backward.  Here it is a LEAF zero tensor:
    # `.grad` receives the rasterizer's gradient buffer itself (no clone), the values are the same zeros, and everything the
    # trainer does with it (reads .grad, slices it, re-assigns it: gs_trainer.py:316-342) works alike.
    # HGS_VIEWSPACE_NONLEAF=1 restores the reference's construction.
    cpp = _frame_call(device)
    if cpp is not None:
        sec = [] if second is None else [second["means3D"], second["feats"], second["opacity"], second["scales"], second["rotations"]]
        image, radii, visible, screenspace_points = cpp.render(
            means3D, feats, opacity, scales, rotations, sec, bg_color, data["world_view_transform"], data["full_proj_transform"],
            data["camera_center"], int(data["image_height"]), int(data["image_width"]), float(data["fovx"]), float(data["fovy"]),
            float(scaling_modifier), int(active_sh_degree))
        return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": visible, "radii": radii}
    n_rows = means3D.shape[0] + (second["means3D"].shape[0] if second is not None else 0)
    if _VIEWSPACE_NONLEAF:
        screenspace_points = torch.zeros(n_rows, 3, dtype=means3D.dtype, requires_grad=True, device=device) + 0
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
    else:
        screenspace_points = torch.zeros(n_rows, 3, dtype=means3D.dtype, requires_grad=True, device=device)

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
    if second is not None:
        second = {"means3D": second["means3D"], "opacities": second["opacity"], "scales": second["scales"],
                  "rotations": second["rotations"], ("colors_precomp" if is_rgb else "shs"): second["feats"]}
    # (visibility = radii > 0 comes back from the rasterizer itself when it is ours: written by the kernel that writes radii)
    out = GaussianRasterizer(raster_settings=settings)(
        means3D=means3D,
        means2D=screenspace_points,
        shs=None if is_rgb else feats,
        colors_precomp=feats if is_rgb else None,
        opacities=opacity,
        scales=scales,
        rotations=rotations,
        clamp_output=True,   # the reference's torch.clamp(rendered_image, 0.0, 1.0) (:153), fused into the blend kernels
        **({} if second is None else {"second": second}),   # (a one-model call carries exactly the reference's kwargs + the clamp)
        **({"with_visibility": True} if _FUSED_VISIBILITY else {}),
    )
    image, radii = out[0], out[1]
    return {
        "render": image,
        "viewspace_points": screenspace_points,
        "visibility_filter": out[2] if len(out) > 2 else radii > 0,
        "radii": radii,
    }


# ---------------------------------------------------------------------------------------------
# Pipelined forward-only rendering for the reference's frame loops.
#
# validate / animate / render_canonical (gs_trainer.py:448-537, 539-586, 588-684) run under torch.no_grad() and call
# render_human_scene once per frame, one frame after the other on one stream; every call waits once on the host for the
# frame's pair count.  Frames are independent, so here they are (a) enqueued without any host wait (deferred frames: the
# binning buffer comes from a persistent, generously sized arena per stream and the count is checked afterwards), and
# (b) dealt round-robin to a few side HIP streams, so that one frame's latency-bound binning runs under another's
# VALU-bound blending.  Results are bit-identical to calling render() / render_human_scene() frame by frame: the same
# kernels run on the same inputs (tests/test_gpu_parity.py::test_render_batch_equals_serial_rendering).
_batch_streams = {}


def _streams_for(device, n):
    pool = _batch_streams.setdefault(device, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device))
    return pool[:n]


def _render_deferred(means3D, feats, opacity, scales, rotations, data, scaling_modifier, bg_color, active_sh_degree):
    device = means3D.device
    if bg_color is None:
        bg_color = torch.zeros(3, dtype=torch.float32, device=device)
    settings = GaussianRasterizationSettings(
        image_height=int(data["image_height"]), image_width=int(data["image_width"]),
        tanfovx=math.tan(data["fovx"] * 0.5), tanfovy=math.tan(data["fovy"] * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=data["world_view_transform"], projmatrix=data["full_proj_transform"],
        sh_degree=active_sh_degree, campos=data["camera_center"], prefiltered=False, debug=False)
    is_rgb = feats.dim() == 2
    return rasterize_deferred(means3D, opacity, settings, shs=None if is_rgb else feats, colors_precomp=feats if is_rgb else None,
                              scales=scales, rotations=rotations, clamp_output=True)


_MAX_PENDING = 256   # deferred frames in flight before the oldest is resolved (the library's result-slot ring has 1 024 entries)


def render_batch(frames, num_streams=2):
    """Forward-only (no autograd graph) rendering of independent frames, pipelined.

    `frames`: an iterable of dicts holding render()'s keyword arguments (means3D, feats, opacity, scales, rotations, data
    [, scaling_modifier, bg_color, active_sh_degree]); it may be a generator that produces each frame's Gaussians on the
    caller's stream as it goes (the posed human of the animation loop).  Returns, in order, what render() returns for
    every frame -- {"render", "viewspace_points", "visibility_filter", "radii"} -- valid on the caller's stream."""
    out, pending = [], []
    main = side = None
    resolved = 0
    with torch.no_grad():
        for i, fr in enumerate(frames):
            # a sliding window: a long frame loop (animation, canonical renders) never has more than _MAX_PENDING unresolved
            # frames -- their result slots in the library are a ring, and an unresolved frame pins its scratch arena's turn
            while len(pending) - resolved >= _MAX_PENDING:
                pending[resolved][0].resolve()
                resolved += 1
            device = fr["means3D"].device
            if main is None:
                main = torch.cuda.current_stream(device)
                side = _streams_for(device, max(1, int(num_streams)))
            st = side[i % len(side)]
            st.wait_stream(main)          # this frame's inputs were produced on the caller's stream
            with torch.cuda.stream(st):
                f = _render_deferred(fr["means3D"], fr["feats"], fr["opacity"], fr["scales"], fr["rotations"], fr["data"],
                                     fr.get("scaling_modifier", 1.0), fr.get("bg_color"), fr.get("active_sh_degree", 0))
            for t in f.keep["inputs"]:    # read by side-stream kernels: keep the caching allocator from recycling them early
                if t is not None and t.is_cuda:
                    t.record_stream(st)
            pending.append((f, fr["means3D"]))
        for f, _ in pending:              # every frame fitted its binning buffer (or is run again, exactly sized)
            f.resolve()
        if main is not None:
            for st in side:
                main.wait_stream(st)
        for f, means3D in pending:
            f.color.record_stream(main)
            f.radii.record_stream(main)
            out.append({"render": f.color, "viewspace_points": torch.zeros_like(means3D), "visibility_filter": f.radii > 0,
                        "radii": f.radii})
    return out


def render_human_scene_batch(items, num_streams=2):
    """render_human_scene for a sequence of frames, forward only, pipelined (see render_batch).  `items`: an iterable of
    dicts with render_human_scene's arguments (data, human_gs_out, scene_gs_out, bg_color [, human_bg_color,
    scaling_modifier, render_mode, render_human_separate]).  Returns the same dicts render_human_scene returns."""
    items = list(items) if not isinstance(items, (list, tuple)) else items
    frames, layout = [], []
    for it in items:
        mode = it.get("render_mode", "human_scene")
        g = _gather(it.get("human_gs_out"), it.get("scene_gs_out"), mode)
        common = {"data": it["data"], "scaling_modifier": it.get("scaling_modifier", 1.0)}
        frames.append({**common, "means3D": g["means3D"], "feats": g["feats"], "opacity": g["opacity"], "scales": g["scales"],
                       "rotations": g["rotations"], "bg_color": it.get("bg_color"), "active_sh_degree": g["active_sh_degree"]})
        separate = bool(it.get("render_human_separate")) and mode == "human_scene"
        if separate:
            h = _gather(it["human_gs_out"], None, "human")
            hbg = it.get("human_bg_color")
            frames.append({**common, "means3D": h["means3D"], "feats": h["feats"], "opacity": h["opacity"], "scales": h["scales"],
                           "rotations": h["rotations"], "bg_color": hbg if hbg is not None else it.get("bg_color"),
                           "active_sh_degree": h["active_sh_degree"]})
        layout.append((mode, separate))
    rendered = iter(render_batch(frames, num_streams))
    out = []
    for it, (mode, separate) in zip(items, layout):
        pkg = next(rendered)
        if separate:
            hp = next(rendered)
            pkg["human_img"], pkg["human_visibility_filter"], pkg["human_radii"] = hp["render"], hp["visibility_filter"], hp["radii"]
        if mode == "human":
            pkg["human_visibility_filter"], pkg["human_radii"] = pkg["visibility_filter"], pkg["radii"]
        elif mode == "human_scene":
            n_h, n_s = it["human_gs_out"]["xyz"].shape[0], it["scene_gs_out"]["xyz"].shape[0]
            pkg["scene_visibility_filter"], pkg["scene_radii"] = pkg["visibility_filter"][n_h:], pkg["radii"][n_h:]
            if "human_visibility_filter" not in pkg:
                pkg["human_visibility_filter"], pkg["human_radii"] = pkg["visibility_filter"][:-n_s], pkg["radii"][:-n_s]
        else:
            pkg["scene_visibility_filter"], pkg["scene_radii"] = pkg["visibility_filter"], pkg["radii"]
        out.append(pkg)
    return out
