"""Where the host time of one forward+backward frame goes, level by level (tiny scene: the GPU work is negligible, so every
figure is host time).  Levels: the C++ node alone (forward under no_grad / forward with a graph / + backward), the
GaussianRasterizer module around it, the renderer adapter (render_human_scene) around that.
HC_P / HC_H / HC_W choose the scene (default 2000 Gaussians at 64x64)."""
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "ml-hugs_amd"))
import diff_gaussian_rasterization as dgr   # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer   # noqa: E402
from hugs_amd import synthetic as syn   # noqa: E402
from hugs_amd.renderer import render_human_scene   # noqa: E402

P, H, W, D = int(os.environ.get("HC_P", 2000)), int(os.environ.get("HC_H", 64)), int(os.environ.get("HC_W", 64)), 0
device = torch.device("cuda", 0)
cam = syn.pinhole_camera(H, W)
g = syn.scene_gaussians(P, cam, seed=0, sigma_px=2.0)
dev = lambda a, grad=False: torch.from_numpy(np.ascontiguousarray(a)).to(device).requires_grad_(grad)
t = {k: dev(g[k], True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
means2D = torch.zeros(P, 3, device=device, requires_grad=True)
dLd = dev(syn.pixel_grad(H, W))
bg = torch.ones(3, device=device)
settings = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam["fovx"] * 0.5),
                                         tanfovy=math.tan(cam["fovy"] * 0.5), bg=bg, scale_modifier=1.0,
                                         viewmatrix=dev(cam["world_view_transform"]), projmatrix=dev(cam["full_proj_transform"]),
                                         sh_degree=D, campos=dev(cam["camera_center"]), prefiltered=False, debug=False)
leaves = list(t.values()) + [means2D]
human = {"xyz": t["means3D"], "scales": t["scales"], "rotq": t["rotations"], "shs": t["shs"], "opacity": t["opacities"],
         "active_sh_degree": D}
data = {"image_height": H, "image_width": W, "fovx": cam["fovx"], "fovy": cam["fovy"],
        "world_view_transform": settings.viewmatrix, "full_proj_transform": settings.projmatrix, "camera_center": settings.campos}
cpp = dgr._load_cpp()
empty = torch.Tensor([])


def cpp_forward():
    return cpp.rasterize(t["means3D"], means2D, t["shs"], empty, t["opacities"], t["scales"], t["rotations"], empty, bg,
                         settings.viewmatrix, settings.projmatrix, settings.campos, H, W, float(settings.tanfovx),
                         float(settings.tanfovy), 1.0, D, False, False, True, [], True)


def drop_grads():
    for x in leaves:
        x.grad = None


def module_forward():
    return GaussianRasterizer(raster_settings=settings)(means3D=t["means3D"], means2D=means2D, opacities=t["opacities"],
                                                        shs=t["shs"], scales=t["scales"], rotations=t["rotations"])


def adapter_forward():
    return render_human_scene(data, human, None, bg_color=bg, render_mode="human")


_lib = dgr._load()
stat = lambda name: _lib.hgs_debug_stat(name.encode())


def timeit(name, fn, n=2000):
    """wall = loop time per call; busy = wall minus the time the library spent spinning for N (the host's only idle time
    inside the loop); lib fwd / bwd = time inside the two C entry points (the forward's without its wait)."""
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
    s0 = {k: stat(k) for k in ("forward_ns", "forward_wait_ns", "backward_ns")}
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    wall = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    d = {k: (stat(k) - s0[k]) / n * 1e-3 for k in s0}
    print(f"{name:62s} wall {wall:6.1f}  busy {wall - d['forward_wait_ns']:6.1f}  lib fwd {d['forward_ns'] - d['forward_wait_ns']:5.1f} "
          f"(+ wait {d['forward_wait_ns']:5.1f})  lib bwd {d['backward_ns']:5.1f}  [us]")
    return wall


def with_no_grad(fn):
    def f():
        with torch.no_grad():
            fn()
    return f


def fwd_bwd(fn, pick):
    def f():
        out = fn()
        pick(out).backward(dLd)
        drop_grads()
    return f


print(f"P={P} {W}x{H} degree {D}; binding: {'C++ node' if cpp is not None else 'ctypes'}")
timeit("noop python call", lambda: None)
timeit("torch.zeros(P,3,requires_grad) (the adapter's viewspace tensor)", lambda: torch.zeros(P, 3, device=device, requires_grad=True))
timeit("torch.empty x5 (what one frame allocates)", lambda: [torch.empty(1 << 16, device=device) for _ in range(5)])
if cpp is not None:
    timeit("C++ node: forward, no_grad", with_no_grad(cpp_forward))
    timeit("C++ node: forward, graph recorded (no backward run)", cpp_forward)
    timeit("C++ node: forward + backward + drop grads", fwd_bwd(cpp_forward, lambda o: o[0]))
timeit("module: forward, no_grad", with_no_grad(module_forward))
timeit("module: forward + backward + drop grads", fwd_bwd(module_forward, lambda o: o[0]))
timeit("adapter: render_human_scene forward, no_grad", with_no_grad(adapter_forward))
timeit("adapter: render_human_scene forward + backward + drop grads", fwd_bwd(adapter_forward, lambda o: o["render"]))
x = torch.zeros(3, H, W, device=device, requires_grad=True)
timeit("reference point: (x * 1).backward(dLd) on a leaf (engine round trip)", lambda: ((x * 1).backward(dLd), setattr(x, "grad", None)))
