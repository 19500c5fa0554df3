"""Host-side cost per fwd+bwd frame: tiny scene so that the GPU work is negligible."""
import cProfile, math, pstats, sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "ml-hugs_amd"))
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from hugs_amd import synthetic as syn
P, H, W, D = int(os.environ.get("HC_P", 2000)), int(os.environ.get("HC_H", 64)), int(os.environ.get("HC_W", 64)), 3
device = torch.device("cuda", 0)
cam = syn.pinhole_camera(H, W)
g = syn.scene_gaussians(P, cam, seed=0, sigma_px=2.0)
dev = lambda a, grad=False: torch.from_numpy(np.ascontiguousarray(a)).to(device).requires_grad_(grad)
t = {k: dev(g[k], True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
means2D = torch.zeros(P, 3, device=device, requires_grad=True)
dLd = dev(syn.pixel_grad(H, W))
settings = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=math.tan(cam["fovx"] * 0.5),
    tanfovy=math.tan(cam["fovy"] * 0.5), bg=torch.ones(3, device=device), scale_modifier=1.0,
    viewmatrix=dev(cam["world_view_transform"]), projmatrix=dev(cam["full_proj_transform"]), sh_degree=D,
    campos=dev(cam["camera_center"]), prefiltered=False, debug=False)
leaves = list(t.values()) + [means2D]
def step(bwd=True):
    rast = GaussianRasterizer(raster_settings=settings)
    color, radii = rast(means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t["shs"],
                        scales=t["scales"], rotations=t["rotations"])
    if bwd:
        color.backward(dLd)
        for x in leaves: x.grad = None
for _ in range(50): step()
torch.cuda.synchronize()
for bwd in (False, True):
    t0 = time.perf_counter()
    for _ in range(500): step(bwd)
    torch.cuda.synchronize()
    print(f"bwd={bwd}: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us per frame (host-bound)")
pr = cProfile.Profile(); pr.enable()
for _ in range(500): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
