"""ctypes front-end of the C oracle (oracle/hgs_oracle.c).

TEST INFRASTRUCTURE ONLY -- may be imported from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never from ml-hugs_amd/.  PARITY UNPINNED at the rasterizer
boundary (see the header of hgs_oracle.c); sub-steps are pinned by tests/golden.

Stages mirror the device pipeline so every intermediate is comparable:
preprocess -> scan -> emit_keys -> sort -> tile_ranges -> blend_forward, and
blend_backward -> preprocess_backward.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force=False):
    names = ["libhgs_oracle_f32.so", "libhgs_oracle_f64.so"]
    if force or not all(os.path.exists(os.path.join(_HERE, n)) for n in names):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


def _lib(dtype):
    dtype = np.dtype(dtype)
    if dtype not in _LIBS:
        build()
        name = "libhgs_oracle_f32.so" if dtype == np.float32 else "libhgs_oracle_f64.so"
        lib = C.CDLL(os.path.join(_HERE, name))
        lib.oracle_scan.restype = C.c_int64
        assert lib.oracle_real_size() == dtype.itemsize
        lib.oracle_set_threads(usable_cpus())   # (OpenMP's own default is every CPU it can see, quota or not)
        _LIBS[dtype] = lib
    return _LIBS[dtype]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _arr(a, dtype, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a), dtype=dtype)
    if shape is not None:
        a = a.reshape(shape)
    return a


def usable_cpus():
    """CPUs this process can actually run on at once: the affinity mask, capped by the cgroup's CPU quota (a container that
    shows 256 CPUs with a 16-CPU quota runs 256 OpenMP threads SLOWER than 16 -- measured on the GPU boxes: 1.70 s against
    0.49 s per frame)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max",):   # cgroup v2: "<quota|max> <period>"
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, -(-int(quota) // int(period))))
        except (OSError, ValueError):
            pass
    try:   # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            n = min(n, max(1, -(-q // per)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def set_threads(n, dtype=np.float32):
    _lib(dtype).oracle_set_threads(int(n))


def set_sort_chunks(n, dtype=np.float32):
    """Tests: cut the radix sort's array into n chunks whatever the thread count (0 = one chunk per thread, the default)."""
    _lib(dtype).oracle_set_sort_chunks(int(n))


def set_upstream_scale_grad(on):
    """True: dL/dscale WITHOUT the scale_modifier factor, as the published kernel computes it (the library's
    HGS_BWD_UPSTREAM_SCALE_GRAD switch); False (default): the true derivative."""
    for d in (np.float32, np.float64):
        _lib(d).oracle_set_upstream_scale_grad(int(bool(on)))


def set_cull_non_pd(on):
    """True: Gaussians whose projected 2-D covariance is not positive definite (det <= 0 or a <= 0: only reachable through a
    non-PSD cov3D_precomp) are culled, as the HIP library does (include/hgs_rasterizer.h); False (default): the published
    algorithm, which culls det == 0 only and blends such a splat wherever its exponent happens to be <= 0."""
    for d in (np.float32, np.float64):
        _lib(d).oracle_set_cull_non_pd(int(bool(on)))


class Inputs:
    """Plain container of rasterizer inputs (numpy). Shapes as SURVEY.md 8a."""

    def __init__(self, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, image_height,
                 image_width, bg, shs=None, colors_precomp=None, scales=None, rotations=None,
                 cov3D_precomp=None, sh_degree=0, scale_modifier=1.0, dtype=np.float32, cull_non_pd=False):
        self.dtype = np.dtype(dtype)
        self.cull_non_pd = bool(cull_non_pd)   # the HIP library's rule for projected covariances that are not positive definite (set_cull_non_pd)
        d = self.dtype
        self.means3D = _arr(means3D, d, (-1, 3))
        self.P = self.means3D.shape[0]
        self.opacities = _arr(opacities, d, (-1,))
        self.shs = _arr(shs, d)
        self.M = 0 if self.shs is None else self.shs.shape[1]
        self.colors_precomp = _arr(colors_precomp, d, (-1, 3))
        self.scales = _arr(scales, d, (-1, 3))
        self.rotations = _arr(rotations, d, (-1, 4))
        self.cov3D_precomp = _arr(cov3D_precomp, d, (-1, 6))
        self.viewmatrix = _arr(viewmatrix, d, (16,))
        self.projmatrix = _arr(projmatrix, d, (16,))
        self.campos = _arr(campos, d, (3,))
        self.bg = _arr(bg, d, (3,))
        # tanfov arrives as a Python double and is narrowed to fp32 at the boundary
        self.tanfovx = float(np.float32(tanfovx)) if d == np.float32 else float(tanfovx)
        self.tanfovy = float(np.float32(tanfovy)) if d == np.float32 else float(tanfovy)
        self.H, self.W = int(image_height), int(image_width)
        self.D = int(sh_degree)
        self.mod = float(scale_modifier)
        assert (self.shs is None) != (self.colors_precomp is None)
        assert (self.cov3D_precomp is None) != (self.scales is None or self.rotations is None) or \
            (self.cov3D_precomp is None and self.scales is not None and self.rotations is not None)

    @property
    def grid(self):
        return (self.W + 15) // 16, (self.H + 15) // 16


def _real(inp):
    return C.c_float if inp.dtype == np.float32 else C.c_double


def forward(inp, stop_after=None):
    """Run the full forward; returns a dict with every intermediate."""
    lib, d, P = _lib(inp.dtype), inp.dtype, inp.P
    r = _real(inp)
    gx, gy = inp.grid
    o = dict(
        depths=np.zeros(P, d), xy=np.zeros((P, 2), d), conic_opacity=np.zeros((P, 4), d),
        rgb=np.zeros((P, 3), d), cov3D=np.zeros((P, 6), d), clamped=np.zeros((P, 3), np.uint8),
        radii=np.zeros(P, np.int32), rect=np.zeros((P, 4), np.int32), tiles_touched=np.zeros(P, np.uint32),
        offsets=np.zeros(P, np.uint32),
    )
    H, W = inp.H, inp.W
    o["color"] = np.zeros((3, H, W), d)
    o["final_T"] = np.ones((H, W), d)
    o["n_contrib"] = np.zeros((H, W), np.uint32)
    o["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    o["N"] = 0
    o["keys"] = np.zeros(0, np.uint64)
    o["values"] = np.zeros(0, np.uint32)
    if P == 0:
        return o  # A.6 quirk 8: colour stays zero, no background
    lib.oracle_set_cull_non_pd(int(getattr(inp, "cull_non_pd", False)))
    try:
        lib.oracle_preprocess(
            C.c_int(P), C.c_int(inp.M), C.c_int(inp.D), C.c_int(H), C.c_int(W), r(inp.tanfovx), r(inp.tanfovy),
            r(inp.mod), _p(inp.means3D), _p(inp.shs), _p(inp.colors_precomp), _p(inp.opacities), _p(inp.scales),
            _p(inp.rotations), _p(inp.cov3D_precomp), _p(inp.viewmatrix), _p(inp.projmatrix), _p(inp.campos),
            _p(o["depths"]), _p(o["xy"]), _p(o["conic_opacity"]), _p(o["rgb"]), _p(o["cov3D"]), _p(o["clamped"]),
            _p(o["radii"]), _p(o["rect"]), _p(o["tiles_touched"]))
    finally:
        lib.oracle_set_cull_non_pd(0)
    if stop_after == "preprocess":
        return o
    N = int(lib.oracle_scan(C.c_int(P), _p(o["tiles_touched"]), _p(o["offsets"])))
    o["N"] = N
    keys = np.zeros(N, np.uint64)
    values = np.zeros(N, np.uint32)
    lib.oracle_emit_keys(C.c_int(P), C.c_int(W), _p(o["depths"]), _p(o["radii"]), _p(o["rect"]),
                         _p(o["offsets"]), _p(keys), _p(values))
    o["keys_unsorted"], o["values_unsorted"] = keys.copy(), values.copy()
    lib.oracle_sort_pairs(C.c_int64(N), _p(keys), _p(values))
    o["keys"], o["values"] = keys, values
    lib.oracle_tile_ranges(C.c_int64(N), _p(keys), C.c_int(gx * gy), _p(o["ranges"]))
    if stop_after == "binning":
        return o
    lib.oracle_blend_forward(C.c_int(H), C.c_int(W), _p(o["ranges"]), _p(values), _p(o["xy"]),
                             _p(o["conic_opacity"]), _p(o["rgb"]), _p(inp.bg), _p(o["color"]),
                             _p(o["final_T"]), _p(o["n_contrib"]))
    return o


def backward(inp, fwd, dL_dcolor_img):
    """Analytic backward given forward intermediates; returns dict of gradients."""
    lib, d, P = _lib(inp.dtype), inp.dtype, inp.P
    r = _real(inp)
    H, W = inp.H, inp.W
    g = dict(
        means2D=np.zeros((P, 3), d), conic=np.zeros((P, 4), d), opacities=np.zeros((P, 1), d),
        colors=np.zeros((P, 3), d), means3D=np.zeros((P, 3), d),
        shs=np.zeros((P, max(inp.M, 1), 3), d), scales=np.zeros((P, 3), d), rotations=np.zeros((P, 4), d),
        cov3D=np.zeros((P, 6), d),
    )
    if P == 0:
        return g
    dpix = _arr(dL_dcolor_img, d, (3, H, W))
    lib.oracle_blend_backward(
        C.c_int(P), C.c_int(H), C.c_int(W), _p(fwd["ranges"]), _p(fwd["values"]), _p(fwd["xy"]),
        _p(fwd["conic_opacity"]), _p(fwd["rgb"]), _p(inp.bg), _p(fwd["final_T"]), _p(fwd["n_contrib"]),
        _p(dpix), _p(g["means2D"]), _p(g["conic"]), _p(g["opacities"]), _p(g["colors"]))
    lib.oracle_preprocess_backward(
        C.c_int(P), C.c_int(inp.M), C.c_int(inp.D), C.c_int(H), C.c_int(W), r(inp.tanfovx), r(inp.tanfovy),
        r(inp.mod), _p(inp.means3D), _p(inp.shs), _p(inp.scales), _p(inp.rotations), _p(inp.cov3D_precomp),
        _p(inp.viewmatrix), _p(inp.projmatrix), _p(inp.campos), _p(fwd["radii"]), _p(fwd["cov3D"]),
        _p(fwd["clamped"]), _p(g["means2D"]), _p(g["conic"]), _p(g["colors"]), _p(g["means3D"]),
        _p(g["shs"]) if inp.shs is not None else None, _p(g["scales"]), _p(g["rotations"]), _p(g["cov3D"]))
    return g


def cov3d(scales, rots, mod=1.0, dtype=np.float32):
    s, q = _arr(scales, dtype, (-1, 3)), _arr(rots, dtype, (-1, 4))
    out = np.zeros((s.shape[0], 6), dtype)
    r = C.c_float if np.dtype(dtype) == np.float32 else C.c_double
    _lib(dtype).oracle_cov3d(C.c_int(s.shape[0]), _p(s), r(mod), _p(q), _p(out))
    return out


def eval_sh(deg, shs, dirs, dtype=np.float32):
    """shs [P,M,3], dirs [P,3] (unit) -> [P,3]; no +0.5, no clamp."""
    s, dd = _arr(shs, dtype), _arr(dirs, dtype, (-1, 3))
    out = np.zeros((s.shape[0], 3), dtype)
    _lib(dtype).oracle_eval_sh(C.c_int(s.shape[0]), C.c_int(s.shape[1]), C.c_int(deg), _p(s), _p(dd), _p(out))
    return out


def mark_visible(means3D, viewmatrix, dtype=np.float32):
    m, v = _arr(means3D, dtype, (-1, 3)), _arr(viewmatrix, dtype, (16,))
    out = np.zeros(m.shape[0], np.uint8)
    _lib(dtype).oracle_mark_visible(C.c_int(m.shape[0]), _p(m), _p(v), _p(out))
    return out.astype(bool)
