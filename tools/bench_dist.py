#!/usr/bin/env python3
"""distCUDA2 (row f-4): the uniform-grid search against the brute-force scan, Gaussian clouds.  One JSON line."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ml-hugs_amd"))
import diff_gaussian_rasterization as dgr      # noqa: E402
from hugs_amd.knn import distCUDA2              # noqa: E402

dev = torch.device("cuda:0")
lib = dgr._load()
lib.hgs_dist_cuda2.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
out = {}
for n in (30_000, 100_000, 300_000, 1_000_000):
    p = torch.randn(n, 3, device=dev)
    res = torch.empty(n, device=dev)

    def timed(fn, iters):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / iters, 3)

    row = {"library_ms": timed(lambda: distCUDA2(p), 5)}
    if n <= 300_000:
        row["brute_force_ms"] = timed(lambda: lib.hgs_dist_cuda2(n, p.data_ptr(), res.data_ptr(), None), 2)
        torch.cuda.synchronize()
        row["same_bits"] = bool(torch.equal(distCUDA2(p).view(torch.int32), res.view(torch.int32)))
    out[str(n)] = row
print(json.dumps({"workload": "distCUDA2, standard-normal clouds; library = grid search from 32 768 points on", **out}))
