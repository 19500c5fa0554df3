#!/usr/bin/env python3
"""Does the library pick the right path for frames nobody tuned it on?  ("next round" item 1.)

The reference renders at the capture's native size (/root/reference/hugs/datasets/neuman.py:346-348), canonical views at 512x512
(/root/reference/hugs/trainer/gs_trainer.py:207-211) and lets the models grow to 524 288 (human) / 2 097 152 (scene) Gaussians
(/root/reference/cfg_files/release/neuman/hugs_human_scene.yaml:89,118).  This tool walks a grid of

    frame sizes 256x256 ... 3840x2160  x  P = 10 000 ... 2 097 152  x  {uniform, trained, human alone at three distances}  x  SH degree {0, 3}

and at every point times forward + backward through the module API with the library's DEFAULT path selection and with every
alternative that has a switch forced (binning mode, frame kind, long-list thresholds, checkpoints, backward form, fused / unfused sort,
emit + scan, deep forward, big-splat groups).  One process: the switches are flipped with hgs_reload_switches().  Timing: the minimum
of `--repeats` loops of `--frames` frames per (point, variant) -- host noise only ever adds time -- with the default re-timed between
the alternatives.  Output: one JSON document -- per point the default's ms, every alternative's ms, the winner and its margin.

    python tools/shape_scan.py --out gpurun_out/shape_scan.json            # the whole grid (~10 min of a box)
    python tools/shape_scan.py --quick                                     # a dozen points
    python tools/shape_scan.py --only 'trained' --sizes 720x1280           # a slice
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-hugs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import diff_gaussian_rasterization as dgr                                   # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, profile_enable, profile_read   # noqa: E402
from hugs_amd import synthetic as syn                                       # noqa: E402
import bench_common as bc                                                   # noqa: E402

SIZES = [(256, 256), (512, 512), (540, 960), (720, 1280), (1080, 1920), (1152, 2048), (2160, 3840)]
COUNTS = [10_000, 30_000, 100_000, 300_000, 1_000_000, 2_097_152]
HUMANS = [6_890, 110_210, 524_288]
HUMAN_DISTS = [3.0, 5.0, 9.0]

# name -> (environment of the library's switches, leave checkpoints?)
VARIANTS = {
    "default": ({}, True),
    "bin_by_cell": ({"HGS_BIN_MODE": "c"}, True),
    "bin_in_order": ({"HGS_BIN_MODE": "o"}, True),
    "kind_sparse": ({"HGS_FRAME_KIND": "s"}, True),
    "kind_dense": ({"HGS_FRAME_KIND": "d"}, True),
    "kind_sparse_no_ckpt": ({"HGS_FRAME_KIND": "s"}, False),
    "kind_dense_no_ckpt": ({"HGS_FRAME_KIND": "d"}, False),
    "long_sparse_256": ({"HGS_LONG_MIN_SPARSE": "256"}, True),
    "long_sparse_1024": ({"HGS_LONG_MIN_SPARSE": "1024"}, True),
    "long_sparse_2048": ({"HGS_LONG_MIN_SPARSE": "2048"}, True),
    "long_dense_768": ({"HGS_LONG_MIN_DENSE": "768"}, True),
    "long_dense_2048": ({"HGS_LONG_MIN_DENSE": "2048"}, True),
    "no_ckpt": ({}, False),
    "bwd_wave_per_tile": ({"HGS_BWD_WAVES_PER_TILE": "1"}, False),
    "bwd_wave_per_quad": ({"HGS_BWD_WAVES_PER_TILE": "4"}, False),
    "bwd_two_launches": ({"HGS_BWD_TWO_LAUNCHES": "1"}, True),
    "unfused_sort_blend": ({"HGS_FUSED_SORT_BLEND": "0"}, True),
    "no_emit_scan": ({"HGS_EMIT_SCAN": "0"}, True),
    "no_deep_forward": ({"HGS_DEEP_FORWARD": "0"}, True),
    # (long-list threshold x depth-parallel forward, the pairs the single switches above do not reach)
    "long_sparse_256_no_deep": ({"HGS_LONG_MIN_SPARSE": "256", "HGS_DEEP_FORWARD": "0"}, True),
    "long_sparse_512": ({"HGS_LONG_MIN_SPARSE": "512"}, True),
    "long_sparse_512_no_deep": ({"HGS_LONG_MIN_SPARSE": "512", "HGS_DEEP_FORWARD": "0"}, True),
    "long_sparse_1024_no_deep": ({"HGS_LONG_MIN_SPARSE": "1024", "HGS_DEEP_FORWARD": "0"}, True),
    "long_sparse_2048_no_deep": ({"HGS_LONG_MIN_SPARSE": "2048", "HGS_DEEP_FORWARD": "0"}, True),
    "deep_min_1536": ({"HGS_DEEP_MIN": "1536"}, True),
    "deep_min_2048": ({"HGS_DEEP_MIN": "2048"}, True),
    "deep_min_3072": ({"HGS_DEEP_MIN": "3072"}, True),
    "long_dense_512": ({"HGS_LONG_MIN_DENSE": "512"}, True),
    "long_dense_768_no_deep": ({"HGS_LONG_MIN_DENSE": "768", "HGS_DEEP_FORWARD": "0"}, True),
    "long_dense_1024": ({"HGS_LONG_MIN_DENSE": "1024"}, True),
    "big_spread": ({"HGS_BIG_PER_GROUP": "0"}, True),
}
SWITCH_NAMES = sorted({k for env, _ in VARIANTS.values() for k in env})


def set_variant(name):
    env, ckpt = VARIANTS[name]
    for k in SWITCH_NAMES:
        os.environ.pop(k, None)
    os.environ.update(env)
    dgr._load().hgs_reload_switches()
    dgr._USE_CKPT = ckpt
    cpp = dgr._load_cpp()
    if cpp is not None:
        cpp.use_checkpoints(ckpt)


def human_gaussians(P, seed=5):
    """tools/bench_c3.py's body: an ellipsoidal blob the size of a person, sizes scaled so that coverage stays that of the SMPL template"""
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((P, 4))
    return {"means3D": (rng.standard_normal((P, 3)) * np.array([0.22, 0.55, 0.14])).astype(np.float32),
            "scales": (0.035 / math.sqrt(P / 6890.0) * np.exp(0.3 * rng.standard_normal((P, 3)))).astype(np.float32),
            "rotations": (q / np.linalg.norm(q, axis=1, keepdims=True) * rng.uniform(0.8, 1.2, (P, 1))).astype(np.float32),
            "shs": (0.3 * rng.standard_normal((P, 16, 3))).astype(np.float32), "opacities": rng.uniform(0.05, 1.0, (P, 1)).astype(np.float32)}


def points(args):
    out = []
    for (H, W) in args.sizes:
        for P in args.counts:
            for kind in ("uniform", "trained"):
                for D in args.degrees:
                    out.append({"kind": kind, "H": H, "W": W, "P": P, "D": D})
    for P in args.humans:
        for dist in HUMAN_DISTS:
            out.append({"kind": "human", "H": 512, "W": 512, "P": P, "D": 0, "dist": dist})
        # the human-only render of a training step is at the capture's size (gs_renderer.py:69)
        out.append({"kind": "human", "H": 720, "W": 1280, "P": P, "D": 0, "dist": 5.0})
        out.append({"kind": "human", "H": 1080, "W": 1920, "P": P, "D": 0, "dist": 5.0})
        out.append({"kind": "human", "H": 512, "W": 512, "P": P, "D": 3, "dist": 5.0})
    if args.tracked:
        # the workloads profiles/collect_round.sh tracks round after round, as frames of this scan: C2 (bench.py), the two frames of C4
        # (tools/bench_c4.py: the joint render and the human-only render at 1080p), the trained-scene profile, C3's two sizes
        out += [{"kind": "tracked_c2", "H": 1080, "W": 1920, "P": 200_000, "D": 3}, {"kind": "tracked_c4_joint", "H": 1080, "W": 1920, "P": 310_210, "D": 0},
                {"kind": "tracked_c4_human", "H": 1080, "W": 1920, "P": 110_210, "D": 0}, {"kind": "tracked_trained", "H": 1080, "W": 1920, "P": 310_210, "D": 0},
                {"kind": "human", "H": 512, "W": 512, "P": 110_210, "D": 0, "dist": 5.0}, {"kind": "human", "H": 512, "W": 512, "P": 6_890, "D": 0, "dist": 5.0},
                # the two renders of tools/bench_step.py (a body SURFACE of 110 210 splats of ~5 pixels, 4 units away: hundreds of lists beyond 2 048 entries)
                {"kind": "tracked_step_joint", "H": 1080, "W": 1920, "P": 310_210, "D": 0}, {"kind": "tracked_step_human", "H": 1080, "W": 1920, "P": 110_210, "D": 0}]
    if args.person_grid:
        # a person on a body surface in front of a covered scene, nearer / farther and denser / sparser: the heavy-tailed frames between
        # a scene render and a human-only one (the depth of the person's lists against the scene's mean)
        for (H, W) in ((1080, 1920), (900, 1600), (720, 1280)):
            for Ps in (200_000, 600_000):
                for Ph in (30_000, 110_210, 300_000):
                    for dist in (3.0, 4.0, 6.0, 9.0):
                        out.append({"kind": "tracked_step_joint", "H": H, "W": W, "P": Ph + Ps, "D": 0, "Ph": Ph, "Ps": Ps, "dist": dist})
    if args.flat_deep:
        # covered frames of 7-pixel splats: flat lists as deep as a trained scene's (E = the mean, 800 .. 1 700)
        for (H, W, P) in ((1080, 1920, 1_000_000), (1080, 1920, 1_300_000), (1080, 1920, 2_097_152), (1024, 1536, 1_000_000), (1152, 2048, 1_500_000), (900, 1600, 1_000_000)):
            out.append({"kind": "uniform7", "H": H, "W": W, "P": P, "D": 0})
    if args.only:
        out = [p for p in out if args.only in p["kind"]]
    if args.where:   # e.g. --where "H==720 and P==630000 and dist==4.0"
        out = [p for p in out if eval(args.where, {}, dict({"dist": None, "Ph": None, "Ps": None}, **p))]
    return out


def build(pt, dev):
    H, W, P, D = pt["H"], pt["W"], pt["P"], pt["D"]
    if pt["kind"] == "human":
        cam = syn.rotating_camera(3, 10, dist=pt["dist"], fov=0.4, img_size=max(H, W))
        if H != W:   # the same rig at the capture's aspect: fov of the longer side
            cam = syn.camera_from_w2c(np.ascontiguousarray(cam["world_view_transform"].T), 0.4, 2.0 * math.atan(math.tan(0.2) * H / W), H, W)
        g = human_gaussians(P)
    elif pt["kind"].startswith("tracked_step"):
        from bench_knn import body_surface
        cam = syn.pinhole_camera(H, W)
        r = np.random.default_rng(3)
        Ph, Ps, dist = pt.get("Ph", 110_210), pt.get("Ps", 200_000), pt.get("dist", 4.0)
        body_surface(6890, r)   # (tools/bench_step.py draws the template first: the same body)
        canon = body_surface(Ph, r, noise=0.004) * 0.6
        q = r.standard_normal((Ph, 4))
        g = {"means3D": (canon + np.array([0.0, 0.0, dist])).astype(np.float32), "rotations": (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32),
             "scales": (0.012 * np.exp(0.3 * r.standard_normal((Ph, 3)))).astype(np.float32), "shs": (0.3 * r.standard_normal((Ph, 16, 3))).astype(np.float32),
             "opacities": r.uniform(0.05, 1.0, (Ph, 1)).astype(np.float32)}
        if pt["kind"] == "tracked_step_joint":
            sg = syn.scene_gaussians(Ps, cam, seed=8, sigma_px=4.0)
            g = {k: np.concatenate([g[k], sg[k]], 0) for k in g}
    elif pt["kind"].startswith("tracked_c4"):
        cam = syn.pinhole_camera(H, W)
        rng = np.random.default_rng(7)
        Ph, Ps, dist = pt.get("Ph", 110_210), pt.get("Ps", 200_000), pt.get("dist", 4.0)
        hm = {"means3D": (rng.standard_normal((Ph, 3)) * np.array([0.22, 0.55, 0.14]) + np.array([0, 0, 4.0])).astype(np.float32),
              "scales": (0.035 / math.sqrt(Ph / 6890.0) * np.exp(0.3 * rng.standard_normal((Ph, 3)))).astype(np.float32),
              "shs": (0.3 * rng.standard_normal((Ph, 16, 3))).astype(np.float32), "opacities": rng.uniform(0.05, 1.0, (Ph, 1)).astype(np.float32)}
        q = rng.standard_normal((Ph, 4))   # (the draws in tools/bench_c4.py's order: the same Gaussians)
        hm["rotations"] = (q / np.linalg.norm(q, axis=1, keepdims=True) * rng.uniform(0.8, 1.2, (Ph, 1))).astype(np.float32)
        g = hm
        if pt["kind"] == "tracked_c4_joint":
            sg = syn.scene_gaussians(Ps, cam, seed=8, sigma_px=4.0)
            g = {k: np.concatenate([hm[k], sg[k]], 0) for k in hm}
    else:
        cam = syn.pinhole_camera(H, W)
        if pt["kind"] == "tracked_trained":
            g = syn.trained_scene_gaussians(200_000, cam, seed=0)
        elif pt["kind"] == "trained":
            Ph = min(110_210, P // 2)
            g = syn.trained_scene_gaussians(P - Ph, cam, seed=0, human=Ph)
        else:
            g = syn.scene_gaussians(P, cam, seed=0, sigma_px=7.0 if pt["kind"] == "uniform7" else 4.0)   # ("uniform7": 7-pixel splats, deep flat lists)
    t = lambda a, grad=False: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev).requires_grad_(grad)
    tens = {k: t(g[k], True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    st = GaussianRasterizationSettings(H, W, math.tan(cam["fovx"] * 0.5), math.tan(cam["fovy"] * 0.5), torch.ones(3, device=dev), 1.0,
                                       t(cam["world_view_transform"]), t(cam["full_proj_transform"]), D, t(cam["camera_center"]), False, False)
    Pn = tens["means3D"].shape[0]
    means2D = torch.zeros(Pn, 3, device=dev, requires_grad=True)
    dL = t(syn.pixel_grad(H, W))
    leaves = list(tens.values()) + [means2D]

    def step():
        color, radii = GaussianRasterizer(st)(means3D=tens["means3D"], means2D=means2D, opacities=tens["opacities"], shs=tens["shs"],
                                              scales=tens["scales"], rotations=tens["rotations"])
        color.backward(dL)
        for x in leaves:
            x.grad = None
        return radii

    step.stats = lambda: list_stats(tens, st)
    return step


def list_stats(tens, st):
    """what the scan kernel knows about the frame when it decides: tiles, non-empty tiles, N, the longest list, lists beyond 256 / 512 /
    1 024 / 2 048 entries (one frame through the ctypes binding's introspection)"""
    _c, _r, h = dgr._debug_forward_state(tens["means3D"].detach(), tens["opacities"].detach(), st, shs=tens["shs"].detach(),
                                         scales=tens["scales"].detach(), rotations=tens["rotations"].detach())
    r = h["ranges"].long()
    lens = (r[:, 1] - r[:, 0]).clamp(min=0)
    ne = lens[lens > 0]
    q = (lambda p: int(torch.quantile(ne.float(), p))) if ne.numel() and ne.numel() < (1 << 24) else (lambda p: None)
    return {"tiles": int(lens.numel()), "nonempty": int(ne.numel()), "N": int(lens.sum()), "longest": int(lens.max()) if lens.numel() else 0,
            "median": q(0.5), "p90": q(0.9), "p99": q(0.99), "sum_sq_over_N": round(float((lens.double() ** 2).sum() / max(1, int(lens.sum()))), 1),
            "beyond": {str(t): int((lens > t).sum()) for t in (256, 512, 768, 1024, 2048, 4096)}}


def time_variant(step, frames, repeats, warm):
    for _ in range(warm):
        step()
    best = float("inf")
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / frames * 1e3)
    return best


def scan_point(pt, args, dev, variants):
    step = build(pt, dev)
    set_variant("default")
    radii = step()
    torch.cuda.synchronize()
    # frames that would take long: fewer of them
    probe = time_variant(step, 3, 1, 3)
    frames = max(6, min(args.frames, int(60.0 / max(probe, 0.02))))
    res = {}
    d0 = time_variant(step, frames, args.repeats, args.warm)
    N, cap, has_long, sparse = bc.last_frame()
    cpp = dgr._load_cpp()
    ck_bytes, ck_used = cpp.last_ckpt_info() if cpp is not None else (None, None)
    profile_enable()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    stages = {k: round(v[0] / 5.0, 4) for k, v in profile_read().items()}
    profile_enable(())
    lists = step.stats() if args.list_stats else None
    defaults = [d0]
    stage_of = {}
    for k, name in enumerate(v for v in variants if v != "default"):
        set_variant(name)
        try:
            res[name] = round(time_variant(step, frames, args.repeats, args.warm), 5)
            if args.stages_of and any(name.startswith(p) for p in args.stages_of):
                profile_enable()
                for _ in range(4):
                    step()
                torch.cuda.synchronize()
                stage_of[name] = {k2: round(v[0] / 4.0, 4) for k2, v in profile_read().items()}
                profile_enable(())
        except RuntimeError as e:   # (a forced path that does not apply to the frame)
            res[name] = None
            print(f"   {name}: {e}", file=sys.stderr)
        if k % 4 == 3:
            set_variant("default")
            defaults.append(time_variant(step, frames, args.repeats, args.warm))
    set_variant("default")
    defaults.append(time_variant(step, frames, args.repeats, args.warm))
    d = min(defaults)
    timed = {k: v for k, v in res.items() if v is not None}
    best = min(timed, key=timed.get) if timed else None
    row = dict(pt)
    row.update({"gaussians": int(radii.numel()), "visible": int((radii > 0).sum()), "num_rendered_N": N, "sparse_frame": sparse, "has_long_tiles": has_long,
                "tiles": ((pt["H"] + 15) // 16) * ((pt["W"] + 15) // 16), "ckpt_MB": None if ck_bytes is None else round(ck_bytes / 1e6, 1),
                "ckpt_slots_used": ck_used, "frames_per_loop": frames, "default_ms": round(d, 5), "default_spread": round(max(defaults) / d, 4),
                "stages_ms": stages, "lists": lists, "forced_stages_ms": stage_of, "forced_ms": res, "best_forced": best, "best_forced_ms": timed.get(best),
                "gain_of_best_forced": None if best is None else round(d / timed[best] - 1.0, 4)})
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--sizes", default=None, help="comma-separated HxW")
    ap.add_argument("--counts", default=None)
    ap.add_argument("--humans", default=None)
    ap.add_argument("--degrees", default="0,3")
    ap.add_argument("--only", default=None)
    ap.add_argument("--variants", default=None, help="comma-separated subset of the forced alternatives")
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--warm", type=int, default=6)
    ap.add_argument("--budget-s", type=float, default=1e9, help="stop opening new points after this many seconds")
    ap.add_argument("--tracked", action="store_true", help="add the tracked workloads (C2, C4's two frames, the trained profile, C3) as points")
    ap.add_argument("--person-grid", action="store_true", help="add a grid of person-in-front-of-a-scene frames (body surface, tools/bench_step.py's geometry)")
    ap.add_argument("--flat-deep", action="store_true", help="add covered frames of 7-pixel splats (flat lists of 800 .. 1 700 entries)")
    ap.add_argument("--where", default=None, help="a Python expression over a point's fields that selects points")
    ap.add_argument("--list-stats", action="store_true", help="record what the scan kernel sees of each frame (non-empty tiles, longest list, ...)")
    ap.add_argument("--stages-of", default="", help="comma-separated prefixes of forced alternatives whose per-stage times are recorded too")
    args = ap.parse_args()
    args.sizes = [tuple(int(x) for x in s.split("x")) for s in args.sizes.split(",")] if args.sizes else SIZES
    args.counts = [int(x) for x in args.counts.split(",")] if args.counts else COUNTS
    args.humans = [int(x) for x in args.humans.split(",")] if args.humans is not None and args.humans != "" else ([] if args.humans == "" else HUMANS)
    args.degrees = [int(x) for x in args.degrees.split(",")]
    args.stages_of = [x for x in args.stages_of.split(",") if x]
    if args.quick:
        args.sizes, args.counts, args.humans, args.degrees = [(512, 512), (720, 1280), (1080, 1920)], [30_000, 300_000], [110_210], [0]
    if args.sizes == [(0, 0)]:
        args.sizes = []
    variants = ["default"] + [v for v in (args.variants.split(",") if args.variants else VARIANTS) if v != "default"]
    bc.gpu_unique_id()   # (asked before this process touches the GPU: bench_common.py)
    dev = torch.device("cuda:0")
    pts = points(args)
    t_start = time.perf_counter()
    rows = []
    for k, pt in enumerate(pts):
        if time.perf_counter() - t_start > args.budget_s:
            print(f"budget spent after {k} of {len(pts)} points", file=sys.stderr)
            break
        try:
            row = scan_point(pt, args, dev, variants)
        except RuntimeError as e:
            row = dict(pt)
            row["error"] = str(e)[:300]
            set_variant("default")
        rows.append(row)
        g = row.get("gain_of_best_forced")
        print(f"[{k + 1}/{len(pts)}] {pt['kind']:8s} {pt['W']}x{pt['H']} P={pt['P']} D={pt['D']}" + (f" dist={pt['dist']}" if "dist" in pt else "") + (f" Ph={pt['Ph']}" if "Ph" in pt else "") +
              (f": default {row['default_ms']:.4f} ms, best forced {row['best_forced']} {row['best_forced_ms']:.4f} ms ({100 * g:+.1f} %)"
               f" sparse={row['sparse_frame']} long={row['has_long_tiles']} N={row['num_rendered_N']}" if g is not None else f": {row.get('error')}"),
              file=sys.stderr, flush=True)
        torch.cuda.empty_cache()
        if args.out:   # (a box can go away mid-scan: keep what there is)
            json.dump({"partial": True, "points": rows}, open(args.out + ".partial", "w"))
    worst = sorted((r for r in rows if r.get("gain_of_best_forced") is not None), key=lambda r: -r["gain_of_best_forced"])
    doc = {"what": "forward+backward ms per frame, module API, default path selection against every forced alternative (min over loops)",
           "box": bc.box(), "lib_csrc_sha16": lib_hash(), "variants": {k: v[0] | ({} if v[1] else {"checkpoints": "off"}) for k, v in VARIANTS.items() if k in variants},
           "n_points": len(rows), "n_points_where_forced_wins_by_5pct": sum(1 for r in worst if r["gain_of_best_forced"] > 0.05),
           "worst": [{k: r[k] for k in ("kind", "H", "W", "P", "D", "default_ms", "best_forced", "best_forced_ms", "gain_of_best_forced")} | ({"dist": r["dist"]} if "dist" in r else {})
                     for r in worst[:25]], "points": rows}
    text = json.dumps(doc, indent=1)
    if args.out:
        open(args.out, "w").write(text)
        try:
            os.remove(args.out + ".partial")
        except OSError:
            pass
    else:
        print(text)


def lib_hash():
    try:
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        import build_id
        return build_id.csrc_sha16()
    except Exception:
        return None


if __name__ == "__main__":
    main()
