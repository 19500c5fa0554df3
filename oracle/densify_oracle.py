"""CPU restatement of the densification statistics update (TEST INFRASTRUCTURE ONLY).
Follows /root/reference/hugs/trainer/gs_trainer.py:407-411 and /root/reference/hugs/models/scene.py:460-462;
pinned against those very lines executed from /root/reference by tests/golden/make_golden.py."""
import numpy as np


def update(max_radii2D, xyz_gradient_accum, denom, viewspace_grad, visibility_filter, radii):
    n = visibility_filter.shape[0]
    vis = visibility_filter.astype(bool)
    m, a, d = max_radii2D.copy(), xyz_gradient_accum.copy(), denom.copy()
    m[vis] = np.maximum(m[vis], radii[vis].astype(np.float32))
    g = viewspace_grad[:n][vis, :2].astype(np.float32)
    a[vis] += np.sqrt((g * g).sum(-1, dtype=np.float32), dtype=np.float32).reshape(-1, *a.shape[1:])
    d[vis] += 1
    return m, a, d
