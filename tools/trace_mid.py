#!/usr/bin/env python3
"""Where the long tiles' sort kernel (tile_sort_mid_kernel) spends its time on a C3 frame, item by item (A/B builds with -DHGS_TRACE
only: `tools/ab_build.sh trace -DHGS_TRACE`, then
`HGS_RASTERIZER_LIB=scratch/lib_trace.so HGS_BINDING=ctypes python tools/trace_mid.py [P]`).  Thread 0 of every workgroup stamps the
100 MHz wall clock at its start, when it knows its list, when the keys are in registers, after the bucket sort, after the sorted list
is written and after the compacted lists are (each stamp behind an s_waitcnt 0: the stores have been acknowledged).  Round 5's
reading, DESIGN_HISTORY.md: the kernel lasts as long as its longest list (3 414 entries: 1.4 us of dependent loads, 0.8 keys, 5.0
bucket sort, 0.8 list, 6.6 compacted lists), and ~3 us pass between its last workgroup's end and the next kernel's first start."""
import ctypes, math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-hugs_amd"))
import diff_gaussian_rasterization as dgr
from hugs_amd import synthetic as syn
from hugs_amd.renderer import render_human_scene
P = int(sys.argv[1]) if len(sys.argv) > 1 else 110210
dev = torch.device("cuda:0"); lib = dgr._load()
rng = np.random.default_rng(5)
q = rng.standard_normal((P, 4))
m = {"xyz": (rng.standard_normal((P, 3)) * np.array([0.22, 0.55, 0.14])).astype(np.float32),
     "scales": (0.035 / math.sqrt(P / 6890.0) * np.exp(0.3 * rng.standard_normal((P, 3)))).astype(np.float32),
     "rotq": (q / np.linalg.norm(q, axis=1, keepdims=True) * rng.uniform(0.8, 1.2, (P, 1))).astype(np.float32),
     "shs": (0.3 * rng.standard_normal((P, 16, 3))).astype(np.float32), "opacity": rng.uniform(0.05, 1.0, (P, 1)).astype(np.float32)}
t = lambda a, grad=False: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev).requires_grad_(grad)
human = {k: t(v, True) for k, v in m.items()}; human["active_sh_degree"] = 0
cam = syn.rotating_camera(3, 10, dist=5.0, fov=0.4, img_size=512)
data = {k: (t(v) if isinstance(v, np.ndarray) else v) for k, v in cam.items()}
bg = torch.ones(3, device=dev)
for _ in range(5): render_human_scene(data, human, None, bg_color=bg, render_mode="human")
torch.cuda.synchronize()
rows = 8192
buf = torch.zeros(rows * 8, dtype=torch.int64, device=dev)
lib.hgs_debug_set_trace.argtypes = [ctypes.c_void_p]
assert lib.hgs_debug_set_trace(buf.data_ptr()) == 0
render_human_scene(data, human, None, bg_color=bg, render_mode="human")
torch.cuda.synchronize(); lib.hgs_debug_set_trace(None)
r = buf.cpu().numpy().reshape(rows, 8)
mid = r[6000:6512]; fused = r[:3080]
ran = mid[:, 0] > 0
t0 = mid[ran, 0].min()
print("mid WGs started", ran.sum(), "with item", (mid[:, 1] > 0).sum())
it = mid[mid[:, 5] > 0]
us = lambda x: (x - t0) / 100.0
print("first start 0.0, last start %.2f; last end %.2f us" % (us(mid[ran, 0].max()), us(it[:, 5].max())))
o = np.argsort(-it[:, 7])
print("   n   start  ranges  keys  bucketed  listed  compacted(end)  [us from kernel's first start]")
for k in list(o[:6]) + list(o[len(o)//2:len(o)//2+3]) + list(o[-3:]):
    x = it[k]; print("%5d  %5.2f  %5.2f  %5.2f  %5.2f  %5.2f  %5.2f" % (x[7], us(x[0]), us(x[1]), us(x[2]), us(x[3]), us(x[4]), us(x[5])))
fr = fused[fused[:, 0] > 0]
print("fused kernel first start %.2f us after mid's first start" % us(fr[:, 0].min()))
