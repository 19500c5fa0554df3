"""Drop-in replacement for the `diff_gaussian_rasterization` module the reference imports at
/root/reference/hugs/renderer/gs_renderer.py:11-14 and uses at :126-152.

Same two public names, same field order / kwargs / return values / error behaviour as the
2023 upstream API the reference was written against (12-field settings tuple, 2-tuple return):

    GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier,
                                  viewmatrix, projmatrix, sh_degree, campos, prefiltered, debug)
    GaussianRasterizer(raster_settings).forward(means3D, means2D, opacities, shs=None,
        colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None) -> (color, radii)
    GaussianRasterizer.markVisible(positions) -> bool[P]

All arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI of
include/hgs_rasterizer.h (libhgs_rasterizer.so).  Two bindings of that ABI live here: a C++
autograd node (lib/_hgs_torch.so, the default when built) and a ctypes autograd.Function
(HGS_BINDING=ctypes, and everything outside the per-frame path: deferred frames, profiling,
markVisible).  There is NO CPU or PyTorch fallback: if the library is missing or a tensor is not
on the GPU this module raises.  PyTorch is used only for device memory, streams and autograd plumbing.
"""
import ctypes as C
import os
import threading
from typing import NamedTuple

import torch
import torch.nn as nn

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "rasterize_deferred", "DeferredFrame",
           "library_path", "set_upstream_scale_grad"]

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# HGS_RASTERIZER_LIB lets a test/benchmark point at another build of the same library (A/B runs)
_LIB_PATH = os.environ.get("HGS_RASTERIZER_LIB") or os.path.join(os.path.dirname(_PKG_DIR), "lib", "libhgs_rasterizer.so")
_ABI_VERSION = 11


def library_path():
    return _LIB_PATH


# ---------------------------------------------------------------------------------------------
# ctypes mirror of include/hgs_rasterizer.h
class _Settings(C.Structure):
    _fields_ = [("image_height", C.c_int32), ("image_width", C.c_int32), ("tanfovx", C.c_float),
                ("tanfovy", C.c_float), ("bg", C.c_void_p), ("scale_modifier", C.c_float),
                ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("sh_degree", C.c_int32),
                ("campos", C.c_void_p), ("prefiltered", C.c_int32), ("debug", C.c_int32)]


class _Segment(C.Structure):
    _fields_ = [("P", C.c_int32), ("M", C.c_int32), ("means3D", C.c_void_p), ("shs", C.c_void_p), ("colors_precomp", C.c_void_p),
                ("opacities", C.c_void_p), ("scales", C.c_void_p), ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p)]


class _ForwardArgs(C.Structure):
    _fields_ = [("s", _Settings), ("P", C.c_int32), ("M", C.c_int32), ("means3D", C.c_void_p),
                ("shs", C.c_void_p), ("colors_precomp", C.c_void_p), ("opacities", C.c_void_p),
                ("scales", C.c_void_p), ("rotations", C.c_void_p), ("cov3D_precomp", C.c_void_p),
                ("out_color", C.c_void_p), ("radii", C.c_void_p), ("binning_capacity_hint", C.c_int64),
                ("grad_accum_to_zero", C.c_void_p), ("clamp_output", C.c_int32), ("expect_no_long_tiles", C.c_int32),
                ("defer_n", C.c_int32), ("backward_checkpoints", C.c_int32), ("scratch", C.c_void_p * 4),
                ("scratch_bytes", C.c_size_t * 4), ("seg2", _Segment), ("visible", C.c_void_p), ("ckpt_slots_hint", C.c_int64),
                ("before_wait", C.c_void_p), ("before_wait_ctx", C.c_void_p)]


class _ForwardState(C.Structure):
    _fields_ = [("geom", C.c_void_p), ("geom_bytes", C.c_size_t), ("binning", C.c_void_p),
                ("binning_bytes", C.c_size_t), ("image", C.c_void_p), ("image_bytes", C.c_size_t),
                ("ckpt", C.c_void_p), ("ckpt_bytes", C.c_size_t), ("num_rendered", C.c_int64), ("binning_capacity", C.c_int64), ("sparse_frame", C.c_int32),
                ("has_long_tiles", C.c_int32), ("n_token", C.c_uint64), ("ckpt_slots", C.c_int64), ("ckpt_slots_used", C.c_int64)]


class _BackwardArgs(C.Structure):
    _fields_ = [("fwd", _ForwardArgs), ("state", _ForwardState), ("dL_dout_color", C.c_void_p),
                ("grad_accum", C.c_void_p), ("dL_dmeans2D", C.c_void_p), ("dL_dopacity", C.c_void_p),
                ("dL_dcolors", C.c_void_p), ("dL_dmeans3D", C.c_void_p), ("dL_dcov3D", C.c_void_p),
                ("dL_dsh", C.c_void_p), ("dL_dscales", C.c_void_p), ("dL_drotations", C.c_void_p),
                ("seg2_dL_dopacity", C.c_void_p), ("seg2_dL_dcolors", C.c_void_p), ("seg2_dL_dmeans3D", C.c_void_p),
                ("seg2_dL_dcov3D", C.c_void_p), ("seg2_dL_dsh", C.c_void_p), ("seg2_dL_dscales", C.c_void_p),
                ("seg2_dL_drotations", C.c_void_p), ("flags", C.c_uint32), ("reserved", C.c_uint32),
                ("add_dL_dopacity", C.c_void_p), ("add_dL_dcolors", C.c_void_p), ("add_dL_dmeans3D", C.c_void_p),
                ("add_dL_dcov3D", C.c_void_p), ("add_dL_dsh", C.c_void_p), ("add_dL_dscales", C.c_void_p),
                ("add_dL_drotations", C.c_void_p), ("wait_before_per_gaussian", C.c_void_p)]


_ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int, C.c_size_t)
HGS_BWD_UPSTREAM_SCALE_GRAD = 1
# dL/dscales: the true derivative (includes settings.scale_modifier) by default; True reproduces the published kernel, which
# omits the factor (both bindings; HGS_UPSTREAM_SCALE_GRAD=1 in the environment, or set_upstream_scale_grad()).  The two
# agree wherever the reference differentiates: it renders with scale_modifier 1.0 (gs_renderer.py:26,103).
_UPSTREAM_SCALE_GRAD = os.environ.get("HGS_UPSTREAM_SCALE_GRAD", "0") == "1"


def set_upstream_scale_grad(on):
    global _UPSTREAM_SCALE_GRAD
    _UPSTREAM_SCALE_GRAD = bool(on)
    if _cpp is not None:
        _cpp.set_upstream_scale_grad(bool(on))
_lib = None


def _load():
    """Load the HIP library; fail loudly (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"diff_gaussian_rasterization (MI355X): {_LIB_PATH} not found. Build it with "
            "`make -C ml-hugs_amd/csrc` (or __graft_entry__.build()). There is no CPU fallback.")
    lib = C.CDLL(_LIB_PATH)
    lib.hgs_abi_version.restype = C.c_int32
    if lib.hgs_abi_version() != _ABI_VERSION:
        raise RuntimeError("libhgs_rasterizer.so ABI version mismatch; rebuild it")
    lib.hgs_rasterize_forward.restype = C.c_int64
    lib.hgs_rasterize_forward.argtypes = [C.POINTER(_ForwardArgs), _ALLOC_FN, C.c_void_p,
                                          C.POINTER(_ForwardState), C.c_void_p]
    lib.hgs_rasterize_backward.restype = C.c_int32
    lib.hgs_rasterize_backward.argtypes = [C.POINTER(_BackwardArgs), C.c_void_p]
    lib.hgs_forward_poll.restype = C.c_int64
    lib.hgs_forward_poll.argtypes = [C.POINTER(_ForwardState), C.c_int32, C.c_void_p]
    lib.hgs_mark_visible.restype = C.c_int32
    lib.hgs_mark_visible.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hgs_last_error.restype = C.c_char_p
    for fn in (lib.hgs_geom_bytes, lib.hgs_image_bytes, lib.hgs_binning_bytes, lib.hgs_ckpt_bytes, lib.hgs_ckpt_bytes_for_slots, lib.hgs_scratch_offset):
        fn.restype = C.c_size_t
    lib.hgs_ckpt_bytes_for_slots.argtypes = [C.c_int64]
    lib.hgs_ckpt_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    lib.hgs_geom_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.hgs_image_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.hgs_binning_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    lib.hgs_scratch_offset.argtypes = [C.c_char_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32]
    lib.hgs_debug_stat.argtypes = [C.c_char_p]
    lib.hgs_debug_stat.restype = C.c_int64
    lib.hgs_reload_switches.argtypes = []
    lib.hgs_reload_switches.restype = None
    lib.hgs_copy_bandwidth.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.hgs_copy_bandwidth.restype = C.c_int32
    lib.hgs_profile_enable.argtypes = [C.c_uint32]
    lib.hgs_profile_enable.restype = None
    lib.hgs_profile_reset.restype = None
    lib.hgs_profile_set_sampling.argtypes = [C.c_uint32]
    lib.hgs_profile_set_sampling.restype = None
    lib.hgs_profile_read.argtypes = [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.hgs_profile_read.restype = C.c_int32
    lib.hgs_stage_name.argtypes = [C.c_int32]
    lib.hgs_stage_name.restype = C.c_char_p
    _lib = lib
    return lib


STAGES = ("preprocess", "scan", "emit_keys", "sort", "blend_forward", "blend_backward", "preprocess_backward")


def profile_enable(stages=STAGES, every_nth=1):
    """Time the named stages with HIP events on the launch stream (empty tuple disables); every_nth > 1 samples."""
    lib = _load()
    mask = 0
    for s in stages:
        mask |= 1 << STAGES.index(s)
    lib.hgs_profile_reset()
    lib.hgs_profile_set_sampling(every_nth)
    lib.hgs_profile_enable(mask)


def profile_read():
    """{stage: (total_ms, launches)} since the last profile_enable(); synchronises the events."""
    lib = _load()
    out = {}
    for i, s in enumerate(STAGES):
        ms, n = C.c_double(0), C.c_int64(0)
        lib.hgs_profile_read(i, C.byref(ms), C.byref(n))
        if n.value:
            out[s] = (ms.value, n.value)
    return out


def _raise_last(lib, what):
    raise RuntimeError(f"{what}: {lib.hgs_last_error().decode()}")


def _ptr(t):
    return None if t is None or t.numel() == 0 else t.data_ptr()


def _f32c(t):
    """contiguous fp32 (upstream calls .contiguous() on every input); None / empty -> None"""
    if t is None or t.numel() == 0:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _require_gpu(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"diff_gaussian_rasterization (MI355X): `{name}` must live on the GPU "
                           "(HIP device); there is no CPU fallback")


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


# ---------------------------------------------------------------------------------------------
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


def _fill_settings(s, rs, keep):
    dev = keep["device"]
    # (the usual case -- fp32, contiguous, already on the device -- costs one attribute check per tensor)
    bg, vm, pm, cp = (x if (x.device == dev and x.dtype == torch.float32 and x.is_contiguous()) else _f32c(x.to(dev))
                      for x in (rs.bg, rs.viewmatrix, rs.projmatrix, rs.campos))
    keep["settings_tensors"] = (bg, vm, pm, cp)
    s.image_height, s.image_width = int(rs.image_height), int(rs.image_width)
    s.tanfovx, s.tanfovy = float(rs.tanfovx), float(rs.tanfovy)
    s.bg, s.viewmatrix, s.projmatrix, s.campos = bg.data_ptr(), vm.data_ptr(), pm.data_ptr(), cp.data_ptr()
    s.scale_modifier = float(rs.scale_modifier)
    s.sh_degree = int(rs.sh_degree)
    s.prefiltered = int(bool(rs.prefiltered))
    s.debug = int(bool(rs.debug))


def _fill_forward(a, rs, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, keep):
    _fill_settings(a.s, rs, keep)
    a.P = int(means3D.shape[0])
    a.M = int(sh.shape[1]) if sh is not None else 0
    a.means3D, a.shs, a.colors_precomp = _ptr(means3D), _ptr(sh), _ptr(colors_precomp)
    a.opacities, a.scales, a.rotations = _ptr(opacities), _ptr(scales), _ptr(rotations)
    a.cov3D_precomp = _ptr(cov3D_precomp)


# Last frame's N per (device, P, H, W): the next frame of the same shape passes it (plus a margin) as
# binning_capacity_hint, so the library enqueues the whole frame before it waits for this frame's N.
# HGS_BINNING_HINT=0 turns the guess off (forward then waits for N before binning, as upstream does).
_last_num_rendered = {}
_USE_HINT = os.environ.get("HGS_BINNING_HINT", "1") != "0"


_max_num_rendered = {}   # largest N seen per shape key: sizes the binning arena of deferred frames
_DEFERRED_MIN_CAPACITY = 1 << 22   # ceiling of the floor below


def _deferred_capacity(seen, H, W):
    """Binning capacity of a deferred frame: 4x the largest N its shape has shown, and no less than 256 entries per tile
    (a 1080p frame: 2 Mi entries, a 512x512 one: 256 Ki -- the floor scales with the image instead of being 4 Mi flat)."""
    tiles = ((H + 15) // 16) * ((W + 15) // 16)
    return max(4 * seen + 4096, min(_DEFERRED_MIN_CAPACITY, max(1 << 16, 256 * tiles)))


# Whether the last frame of a shape was SPARSE (few non-empty tiles): such frames -- and dense ones that had long tiles -- get a
# checkpoint buffer when a backward will follow (hgs_forward_args.backward_checkpoints); a shape without history is assumed sparse (the library then
# allocates the buffer only if the frame turns out to be).
_last_sparse = {}
_USE_CKPT = os.environ.get("HGS_BWD_SEGMENTED", "1") != "0"


_last_ckpt_used = {}   # checkpoint slots the last frame of a shape needed: sizes the next frame's checkpoint buffer


def _round_ckpt_slots(used):
    """+ 25 % + 64, in steps of 1 024 slots (4 MB); csrc_torch/hgs_torch.cpp round_ckpt_slots computes the same number"""
    want = used + used // 4 + 64
    return (want + 1023) // 1024 * 1024


def _ckpt_hint(key):
    """hgs_forward_args.ckpt_slots_hint from the previous frame of this shape (0: none -- the full layout)"""
    used = _last_ckpt_used.get(key, 0) if _USE_HINT else 0
    return _round_ckpt_slots(used) if used > 0 else 0


_HINT_SHAPES = 256   # shapes remembered (densification changes P all the time: do not grow without bound)


def _remember(key, n, has_long, sparse=None, to_cpp=True):
    # least recently used shapes go first -- not the whole table at once (round 3 wiped all three tables at 256 shapes: a
    # densifying run then paid one blocking frame per live shape each time); dicts keep insertion order, re-inserting = touching
    _last_num_rendered.pop(key, None)
    while len(_last_num_rendered) >= _HINT_SHAPES:
        old = next(iter(_last_num_rendered))
        del _last_num_rendered[old]
        _max_num_rendered.pop(old, None)
        _last_sparse.pop(old, None)
        _last_ckpt_used.pop(old, None)
    if sparse is not None:
        _last_sparse[key] = bool(sparse)
    if to_cpp and _cpp is not None:   # ... and what the Python paths learnt, the C++ node uses
        _cpp.set_hint(key[0], key[1], key[2], key[3], n, has_long, _last_sparse.get(key, True))
    _last_num_rendered[key] = (n, has_long)
    _max_num_rendered[key] = max(n, _max_num_rendered.get(key, 0))


def _capacity_hint(key):
    """(binning_capacity_hint, expect_no_long_tiles) from the previous frame of this shape"""
    prev = _last_num_rendered.get(key) if _USE_HINT else None
    if prev is None:
        return 0, 0
    n, had_long = prev
    return _round_capacity(n), 0 if had_long else 1


def _round_capacity(n):
    """+ 12.5 % + 4096, rounded up to a granule PROPORTIONAL to the size (1/16 of the next power of two, at least 64 Ki
    entries): frame after frame asks the caching allocator for the same size, and a small frame (the 6 890-Gaussian SMPL
    template at 512x512) no longer carries the 256 Ki-entry granule -- 16.8 MB of binning scratch -- of a 1080p scene.
    (The C++ binding computes the same number: csrc_torch/hgs_torch.cpp round_capacity.)"""
    want = n + n // 8 + 4096
    granule = max(1 << 16, (1 << max(want - 1, 1).bit_length()) >> 4)
    return (want + granule - 1) // granule * granule


_GRAD_NAMES = ("grad_accum", "dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
               "dL_drotations", "seg2_dL_dopacity", "seg2_dL_dcolors", "seg2_dL_dmeans3D", "seg2_dL_dcov3D", "seg2_dL_dsh",
               "seg2_dL_dscales", "seg2_dL_drotations")


def _grad_layout(P, M, P2=0, M2=0):
    """Element offsets of the [P + P2,12] atomic accumulator, the joint dL/dmeans2D and the seven gradient tensors of each
    segment inside one fp32 slab."""
    Pt = P + P2
    sizes = (12 * Pt, 3 * Pt, P, 3 * P, 3 * P, 6 * P, 3 * M * P, 3 * P, 4 * P,
             P2, 3 * P2, 3 * P2, 6 * P2, 3 * M2 * P2, 3 * P2, 4 * P2)
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + 63) // 64 * 64
    return sizes, offs, max(total, 1)


def _grad_slab(P, M, dev, zero, bw, P2=0, M2=0):
    """One allocation for the accumulator (the only part that must be zero -- the library overwrites every other element;
    `zero=False` when forward is asked to zero it) and the gradient tensors; the backward argument block is pointed at
    it by address arithmetic, the tensor views are only made for what backward() returns."""
    sizes, offs, total = _grad_layout(P, M, P2, M2)
    slab = torch.empty(total, dtype=torch.float32, device=dev)
    if zero:
        slab[:max(offs[1], 1)].zero_()
    base = slab.data_ptr()
    for k, name in enumerate(_GRAD_NAMES):
        setattr(bw, name, base + 4 * offs[k] if (sizes[k] or k not in (6, 13)) else None)
    return slab


def _grad_view(slab, dims, k, *shape):
    sizes, offs, _ = _grad_layout(*dims)
    return slab[offs[k]:offs[k] + sizes[k]].view(*shape)


def _align(n, a=256):
    return (n + a - 1) // a * a


# Scratch for frames that need no backward (validation / animation / canonical render loops run under torch.no_grad(),
# gs_trainer.py:448-684): one persistent arena per (device, stream) instead of three allocations per frame.  Work on a
# stream is ordered, so the next frame on that stream may overwrite it.
_arenas = {}
_MAX_ARENAS = 32   # (device, stream, host thread) triples that keep a persistent scratch arena


def _arena(dev, stream_id, nbytes):
    # (per host thread as well: two threads issuing frames on ONE stream interleave their enqueues, and the second frame's
    #  first kernel would overwrite scratch the first frame's later kernels have yet to read)
    key = (dev.index, stream_id, threading.get_ident())
    t = _arenas.pop(key, None)   # (re-inserted below: the dict's order is the order of last use)
    if t is None or t.numel() < nbytes:
        t = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=dev)
    _arenas[key] = t
    while len(_arenas) > _MAX_ARENAS:   # threads and streams come and go: the least recently used arenas are let go
        del _arenas[next(iter(_arenas))]
    return t


def _provide_scratch(args, lib, dev, P, H, W, capacity, persistent_for_stream=None, checkpoints=False, ckpt_slots=0):
    """Pre-sized scratch handed to the library through args.scratch (no allocation callbacks): one buffer holding geom |
    image | binning(capacity) [| checkpoints(capacity)].  Returns (buffer, (offsets))."""
    g, im = _align(lib.hgs_geom_bytes(P, H, W)), _align(lib.hgs_image_bytes(H, W))
    b = _align(lib.hgs_binning_bytes(capacity, H, W)) if capacity > 0 else 0
    ck = 0
    if capacity > 0 and checkpoints:
        ck = _align(lib.hgs_ckpt_bytes_for_slots(ckpt_slots) if ckpt_slots > 0 else lib.hgs_ckpt_bytes(capacity, H, W))
    total = g + im + b + ck
    buf = _arena(dev, persistent_for_stream, total) if persistent_for_stream is not None else \
        torch.empty(total, dtype=torch.uint8, device=dev)
    base = buf.data_ptr()
    base_al = _align(base)
    o = base_al - base            # (torch's caching allocator hands out 512-byte aligned blocks: o == 0)
    args.scratch[0], args.scratch_bytes[0] = base_al, g
    args.scratch[2], args.scratch_bytes[2] = base_al + g, im
    if b:
        args.scratch[1], args.scratch_bytes[1] = base_al + g + im, b
    if ck:
        args.scratch[3], args.scratch_bytes[3] = base_al + g + im + b, ck
    return buf, (o, g, o + g, im, o + g + im, b)   # geom off/len, image off/len, binning off/len


def _check_segment(device, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp):
    """The second set of Gaussians goes to the kernels as raw pointers: a tensor on another device, of another dtype or
    with another row count than means3D would be read (or its gradient written) out of bounds instead of raising."""
    P2 = int(means3D.shape[0])
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise RuntimeError("second: means3D must have dimensions (num_points, 3)")
    for name, t, tail in (("means3D", means3D, (3,)), ("shs", sh, None), ("colors_precomp", colors_precomp, (3,)),
                          ("opacities", opacities, None), ("scales", scales, (3,)), ("rotations", rotations, (4,)),
                          ("cov3D_precomp", cov3D_precomp, (6,))):
        if t is None or t.numel() == 0:
            continue
        if t.device != device:
            raise RuntimeError(f"second: {name} is on {t.device}, the first set of Gaussians on {device}")
        if t.dtype != torch.float32:
            raise RuntimeError(f"second: {name} must be float32, got {t.dtype}")
        if int(t.shape[0]) != P2:
            raise RuntimeError(f"second: {name} has {int(t.shape[0])} rows, means3D has {P2}")
        if name == "shs" and (t.ndim != 3 or t.shape[2] != 3):
            raise RuntimeError("second: shs must have dimensions (num_points, M, 3)")
        if name == "opacities" and t.numel() != P2:
            raise RuntimeError("second: opacities must have dimensions (num_points, 1)")
        if tail is not None and tuple(t.shape[1:]) != tail:
            raise RuntimeError(f"second: {name} must have dimensions (num_points, {tail[0]})")


def _fill_segment(seg, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp):
    seg.P = int(means3D.shape[0])
    seg.M = int(sh.shape[1]) if sh is not None else 0
    seg.means3D, seg.shs, seg.colors_precomp = _ptr(means3D), _ptr(sh), _ptr(colors_precomp)
    seg.opacities, seg.scales, seg.rotations, seg.cov3D_precomp = _ptr(opacities), _ptr(scales), _ptr(rotations), _ptr(cov3D_precomp)


class _RasterizeGaussians(torch.autograd.Function):
    """forward(means3D, means2D, sh, colors, opacities, scales, rotations, cov3Ds, settings, clamp_output
               [, means3D_b, sh_b, colors_b, opacities_b, scales_b, rotations_b, cov3Ds_b])
    The optional second set of Gaussians (hgs_segment: the joint human + scene render without torch.cat) is rendered
    behind the first in index order; means2D then has P + P_b rows, as the reference's viewspace tensor of a joint render."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, clamp_output=False, *second):
        # (clamp_output: bit 0 = clamp the image, bit 1 = also return the visibility filter `radii > 0` as a third output)
        with_visibility = bool(int(clamp_output) & 2)
        clamp_output = bool(int(clamp_output) & 1)
        lib = _load()
        _require_gpu(means3D, "means3D")
        dev = means3D.device
        rs = raster_settings
        means3D = _f32c(means3D) if means3D.numel() else means3D.float().reshape(0, 3)
        if means3D.ndim != 2 or means3D.shape[1] != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        sh, colors_precomp, opacities = _f32c(sh), _f32c(colors_precomp), _f32c(opacities)
        scales, rotations, cov3Ds_precomp = _f32c(scales), _f32c(rotations), _f32c(cov3Ds_precomp)
        P1, H, W = means3D.shape[0], int(rs.image_height), int(rs.image_width)
        sec = None
        if second and second[0] is not None and second[0].numel():
            if P1 == 0:
                raise RuntimeError("a second set of Gaussians needs a non-empty first one")
            sec = tuple(_f32c(x) for x in second)
            if sec[0].ndim != 2 or sec[0].shape[1] != 3:
                raise RuntimeError("means3D must have dimensions (num_points, 3)")
        P2 = sec[0].shape[0] if sec is not None else 0
        P = P1 + P2

        # P == 0: nothing is launched and colour stays zero (no background) -- upstream behaviour
        color = torch.zeros(3, H, W, dtype=torch.float32, device=dev) if P == 0 else \
            torch.empty(3, H, W, dtype=torch.float32, device=dev)
        radii = torch.empty(P, dtype=torch.int32, device=dev)
        visible = torch.empty(P, dtype=torch.bool, device=dev) if with_visibility else None

        keep = {"device": dev}
        bufs = {}

        def _alloc(_ctx, which, nbytes):
            t = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
            bufs[which] = t
            return t.data_ptr()

        cb = _ALLOC_FN(_alloc)
        # The backward call's argument block is built here, around the forward's: forward fills `bw.fwd` and
        # `bw.state` in place, and -- when a gradient will be asked for -- the gradient slab is allocated and its
        # accumulator zeroed now, ahead of the rasterizer's kernels, so that backward() itself is one library call.
        bw = _BackwardArgs()
        args, state = bw.fwd, bw.state
        _fill_forward(args, rs, means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, keep)
        if sec is not None:
            _check_segment(dev, *sec)
            _fill_segment(args.seg2, *sec)
        args.out_color, args.radii = color.data_ptr(), _ptr(radii)
        args.visible = _ptr(visible)
        args.clamp_output = 1 if clamp_output else 0
        M, M2 = int(args.M), int(args.seg2.M)
        needs_grad = P > 0 and any(ctx.needs_input_grad)
        slab = None
        if needs_grad:
            slab = _grad_slab(P1, M, dev, False, bw, P2, M2)
            args.grad_accum_to_zero = bw.grad_accum
        hint_key = (dev.index, P, H, W)
        args.binning_capacity_hint, args.expect_no_long_tiles = _capacity_hint(hint_key)
        # (sparse frames, and dense ones with long tiles: their deep tiles go through the segmented backward too)
        # (... unless the shape's last frame said it leaves none: hgs_forward_state.ckpt_slots_used < 0)
        args.backward_checkpoints = 1 if (needs_grad and _USE_CKPT and _last_ckpt_used.get(hint_key, 0) >= 0 and
                                          (_last_sparse.get(hint_key, True) or _last_num_rendered.get(hint_key, (0, False))[1])) else 0
        prev_dev = torch.cuda.current_device()
        if prev_dev != dev.index:
            torch.cuda.set_device(dev)
        try:
            stream = torch.cuda.current_stream(dev).cuda_stream
            scratch = None
            if P > 0:
                # pre-sized scratch, no allocation callbacks: per frame when backward will need it, else the stream's arena
                if args.backward_checkpoints and args.binning_capacity_hint > 0:
                    args.ckpt_slots_hint = _ckpt_hint(hint_key)
                scratch = _provide_scratch(args, lib, dev, P, H, W, int(args.binning_capacity_hint),
                                           None if needs_grad else stream, bool(args.backward_checkpoints), int(args.ckpt_slots_hint))
            n = lib.hgs_rasterize_forward(C.byref(args), cb, None, C.byref(state), C.c_void_p(stream))
        finally:
            if prev_dev != dev.index:
                torch.cuda.set_device(prev_dev)
        if n < 0:
            _raise_last(lib, "rasterize_gaussians")

        global _last_frame_info
        ctx.num_rendered = int(n)
        ctx.binning_capacity = int(state.binning_capacity)
        _last_frame_info = (ctx.num_rendered, ctx.binning_capacity)
        _remember(hint_key, int(n), bool(state.has_long_tiles), bool(state.sparse_frame))
        _last_ckpt_used[hint_key] = int(state.ckpt_slots_used)
        ctx.bw, ctx.slab, ctx.keep, ctx.dims = bw, slab, keep, (P1, M, P2, M2)
        ctx.scratch, ctx.bufs = scratch, bufs   # kept alive for backward (and read by _debug_forward_state)
        empty = torch.empty(0, device=dev)
        e = lambda x: x if x is not None else empty
        ctx.save_for_backward(means3D, e(sh), e(colors_precomp), e(opacities), e(scales), e(rotations), e(cov3Ds_precomp), radii,
                              *([e(x) for x in sec] if sec is not None else []))
        ctx.n_second_inputs = len(second)
        ctx.set_materialize_grads(False)  # no zero-filled int32 "gradient" for radii
        if with_visibility:
            ctx.mark_non_differentiable(radii, visible)
            return color, radii, visible
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii, _grad_visible=None):
        lib = _load()
        # the saved tensors are what `ctx.bw` points into: unpacking them also runs autograd's in-place-modification check
        saved = ctx.saved_tensors
        means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, _radii = saved[:8]
        sec = saved[8:]
        dev = means3D.device
        P1, M, P2, M2 = ctx.dims
        P = P1 + P2
        bw, slab = ctx.bw, ctx.slab
        none_second = (None,) * ctx.n_second_inputs
        if grad_out_color is None:  # colour did not take part in the loss
            ctx.slab = None
            return (None,) * 10 + none_second
        if slab is None:  # first use is prepared by forward; a second backward (retain_graph) gets a fresh slab
            slab = _grad_slab(P1, M, dev, True, bw, P2, M2)
        ctx.slab = None

        if P > 0:
            grad_out_color = _f32c(grad_out_color)
            bw.dL_dout_color = grad_out_color.data_ptr()
            bw.flags = HGS_BWD_UPSTREAM_SCALE_GRAD if _UPSTREAM_SCALE_GRAD else 0
            prev_dev = torch.cuda.current_device()
            if prev_dev != dev.index:
                torch.cuda.set_device(dev)
            try:
                rc = lib.hgs_rasterize_backward(C.byref(bw), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            finally:
                if prev_dev != dev.index:
                    torch.cuda.set_device(prev_dev)
            if rc < 0:
                _raise_last(lib, "rasterize_gaussians_backward")

        v = lambda k, *shape: _grad_view(slab, ctx.dims, k, *shape)
        # order of forward's inputs: means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
        # cov3Ds_precomp, raster_settings, clamp_output [, the second set in the same order without means2D]
        first = (v(4, P1, 3), v(1, P, 3),
                 v(6, P1, M, 3) if sh.numel() else None,
                 v(3, P1, 3) if colors_precomp.numel() else None,
                 v(2, P1, 1),
                 v(7, P1, 3) if scales.numel() else None,
                 v(8, P1, 4) if rotations.numel() else None,
                 v(5, P1, 6) if cov3Ds_precomp.numel() else None,
                 None, None)
        if not sec:
            return first + none_second
        _m3b, sh_b, col_b, _op_b, sc_b, rot_b, cov_b = sec
        return first + (v(11, P2, 3),
                        v(13, P2, M2, 3) if sh_b.numel() else None,
                        v(10, P2, 3) if col_b.numel() else None,
                        v(9, P2, 1),
                        v(14, P2, 3) if sc_b.numel() else None,
                        v(15, P2, 4) if rot_b.numel() else None,
                        v(12, P2, 6) if cov_b.numel() else None)


# The same binding as a C++ autograd node (ml-hugs_amd/csrc_torch/hgs_torch.cpp -> lib/_hgs_torch.so): identical library
# calls and policy without the interpreter in the per-frame path.  Used when it has been built (__graft_entry__.build()
# does); HGS_BINDING=ctypes forces the Python path above.  Both are the HIP path -- neither is a fallback for the kernels.
_cpp = None
_CPP_WANTED = os.environ.get("HGS_BINDING", "cpp") != "ctypes"


def _load_cpp():
    global _cpp, _CPP_WANTED
    if _cpp is not None or not _CPP_WANTED:
        return _cpp
    path = os.path.join(os.path.dirname(_LIB_PATH), "_hgs_torch.so")
    if not os.path.exists(path) or os.environ.get("HGS_RASTERIZER_LIB"):   # (an A/B library build has no matching binding)
        _CPP_WANTED = False
        return None
    _load()   # libhgs_rasterizer.so first (ABI check); the extension links against it
    import importlib.util
    spec = importlib.util.spec_from_file_location("_hgs_torch", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if mod.abi_version() != _ABI_VERSION:
        raise RuntimeError("_hgs_torch.so was built against another ABI version; rebuild it")
    mod.use_hints(_USE_HINT)
    mod.use_checkpoints(_USE_CKPT)
    mod.set_upstream_scale_grad(_UPSTREAM_SCALE_GRAD)
    _cpp = mod
    return mod


_last_frame_info = (None, None)


def last_frame_info():
    """(N, binning capacity) of the last forward, whichever binding ran it."""
    return _last_frame_info


_SECOND_KEYS = ("means3D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp")


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, clamp_output=False, second=None, with_visibility=False):
    """`with_visibility`: also return `radii > 0` (bool [P]) as a third tensor, written by the kernel that writes radii.
    `second`: optional dict with the keys of _SECOND_KEYS (missing / None = absent) -- a second model's Gaussians
    rendered together with the first without concatenating anything (hgs_segment); means2D must then have
    len(means3D) + len(second["means3D"]) rows."""
    global _last_frame_info
    sec = ()
    if second is not None and second.get("means3D") is not None and second["means3D"].numel():
        empty = torch.Tensor([])
        sec = tuple(second.get(k) if second.get(k) is not None else empty for k in _SECOND_KEYS)
    cpp = _load_cpp()
    if cpp is not None:
        rs = raster_settings
        out = cpp.rasterize(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs.bg,
                            rs.viewmatrix, rs.projmatrix, rs.campos, int(rs.image_height), int(rs.image_width),
                            float(rs.tanfovx), float(rs.tanfovy), float(rs.scale_modifier), int(rs.sh_degree),
                            bool(rs.prefiltered), bool(rs.debug), bool(clamp_output), list(sec), bool(with_visibility))
        color, radii, visible = out[0], out[1], (out[2] if with_visibility else None)
        n, cap, has_long, sparse = cpp.last_frame_info()
        _last_frame_info = (n, cap)
        if radii.numel():   # one hint table for both bindings: what the C++ node learnt, the Python paths (deferred frames) use
            _remember((means3D.device.index, radii.numel(), int(rs.image_height), int(rs.image_width)), n, has_long, sparse, to_cpp=False)
        return (color, radii, visible) if with_visibility else (color, radii)
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, (1 if clamp_output else 0) | (2 if with_visibility else 0), *sec)


class DeferredFrame:
    """A forward-only frame that was enqueued without the host waiting for its N (hgs_forward_args.defer_n).  `color`
    and `radii` are valid once resolve() has returned (it checks that the frame fitted its binning buffer and, if it
    did not, runs it again with the exact size -- same output tensors, same stream)."""
    __slots__ = ("color", "radii", "args", "state", "keep", "stream", "key", "num_rendered")

    def resolve(self):
        if self.num_rendered is not None:
            return self.num_rendered
        lib = _load()
        dev = self.color.device
        with torch.cuda.device(dev):
            n = lib.hgs_forward_poll(C.byref(self.state), 1, C.c_void_p(self.stream.cuda_stream))
            # HGS_ERR_OVERFLOW: the gated kernels did nothing; HGS_ERR_EXPIRED: the frame ran but its N was lost to a later
            # frame (more than 1 024 forwards before this resolve) -- either way run the frame again, waiting for N this time
            if n in (-6, -7) and b"2^32" not in lib.hgs_last_error():
                self.args.defer_n, self.args.binning_capacity_hint = 0, 0
                bufs = []

                def _alloc(_ctx, which, nbytes):
                    bufs.append(torch.empty(int(nbytes), dtype=torch.uint8, device=dev))
                    return bufs[-1].data_ptr()

                with torch.cuda.stream(self.stream):
                    n = lib.hgs_rasterize_forward(C.byref(self.args), _ALLOC_FN(_alloc), None, C.byref(self.state),
                                                  C.c_void_p(self.stream.cuda_stream))
                    for b in bufs:
                        b.record_stream(self.stream)
            if n < 0:
                _raise_last(lib, "rasterize_gaussians (deferred frame)")
        self.num_rendered = int(n)
        _remember(self.key, int(n), bool(self.state.has_long_tiles), bool(self.state.sparse_frame))
        return self.num_rendered


def rasterize_deferred(means3D, opacities, raster_settings, shs=None, colors_precomp=None, scales=None, rotations=None,
                       cov3D_precomp=None, clamp_output=False):
    """Forward-only rasterization on torch's CURRENT stream without any host wait: returns a DeferredFrame whose
    .color / .radii the GPU fills asynchronously; call .resolve() before handing them to anyone (hugs_amd.renderer.
    render_batch does).  Scratch comes from the stream's persistent arena, sized for 4x the largest N this shape has
    shown (at least 4 Mi entries).  The first frame of a shape has no history and is an ordinary (waiting) frame."""
    lib = _load()
    _require_gpu(means3D, "means3D")
    dev = means3D.device
    rs = raster_settings
    with torch.no_grad():
        means3D = _f32c(means3D) if means3D.numel() else means3D.float().reshape(0, 3)
        shs, colors_precomp, opacities = _f32c(shs), _f32c(colors_precomp), _f32c(opacities)
        scales, rotations, cov3D_precomp = _f32c(scales), _f32c(rotations), _f32c(cov3D_precomp)
        P, H, W = means3D.shape[0], int(rs.image_height), int(rs.image_width)
        f = DeferredFrame()
        f.color = torch.zeros(3, H, W, dtype=torch.float32, device=dev) if P == 0 else \
            torch.empty(3, H, W, dtype=torch.float32, device=dev)
        f.radii = torch.empty(P, dtype=torch.int32, device=dev)
        f.keep = {"device": dev, "inputs": (means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp)}
        f.args, f.state = _ForwardArgs(), _ForwardState()
        f.stream, f.key, f.num_rendered = torch.cuda.current_stream(dev), (dev.index, P, H, W), None
        _fill_forward(f.args, rs, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, f.keep)
        f.args.out_color, f.args.radii = f.color.data_ptr(), _ptr(f.radii)
        f.args.clamp_output = 1 if clamp_output else 0
        if P == 0:
            f.num_rendered = 0
            return f
        seen = _max_num_rendered.get(f.key)
        if seen is None and _cpp is not None:   # (frames that went through the C++ binding's one-call entry points keep their record there)
            known = _cpp.get_hint(*f.key)
            if known is not None:
                seen = int(known[0])
        bufs = []

        def _alloc(_ctx, which, nbytes):
            bufs.append(torch.empty(int(nbytes), dtype=torch.uint8, device=dev))
            return bufs[-1].data_ptr()

        with torch.cuda.device(dev):
            if seen is None:    # nothing known about this shape yet: an ordinary frame that waits for N
                f.keep["scratch"] = _provide_scratch(f.args, lib, dev, P, H, W, 0, f.stream.cuda_stream)
            else:
                cap = _deferred_capacity(seen, H, W)
                f.args.binning_capacity_hint, f.args.defer_n = cap, 1
                f.keep["scratch"] = _provide_scratch(f.args, lib, dev, P, H, W, cap, f.stream.cuda_stream)
            n = lib.hgs_rasterize_forward(C.byref(f.args), _ALLOC_FN(_alloc), None, C.byref(f.state),
                                          C.c_void_p(f.stream.cuda_stream))
        f.keep["bufs"] = bufs
        if n < 0:
            _raise_last(lib, "rasterize_gaussians (deferred)")
        if seen is None:
            f.num_rendered = int(n)
            _remember(f.key, int(n), bool(f.state.has_long_tiles), bool(f.state.sparse_frame))
    return f


_EMPTY = torch.Tensor([])


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """bool[P]: view-space z > 0.2 (upstream's frustum pre-filter; unused by the reference)."""
        lib = _load()
        _require_gpu(positions, "positions")
        with torch.no_grad():
            pos = _f32c(positions)
            P = 0 if pos is None else pos.shape[0]
            present = torch.zeros(P, dtype=torch.bool, device=positions.device)
            if P:
                vm = _f32c(self.raster_settings.viewmatrix.to(positions.device))
                with torch.cuda.device(positions.device):
                    rc = lib.hgs_mark_visible(P, pos.data_ptr(), vm.data_ptr(), present.data_ptr(),
                                              _stream_ptr(positions.device))
                if rc < 0:
                    _raise_last(lib, "mark_visible")
        return present

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, clamp_output=False, second=None, with_visibility=False):
        """Three additions to upstream's signature (`with_visibility`: a third return value, the bool tensor `radii > 0` that
        the reference's render() computes right after this call, gs_renderer.py:159, written by the kernel that writes radii): `clamp_output` -- True fuses the `torch.clamp(image, 0, 1)` that the
        reference's render() applies right after this call (gs_renderer.py:153), forward and backward -- and `second`, a
        dict (means3D, opacities, shs | colors_precomp, scales + rotations | cov3D_precomp) with a second model's Gaussians,
        rendered behind the first in index order exactly as if the tensors had been concatenated (gs_renderer.py:33-37)
        but read, and their gradients written, in place."""
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = _EMPTY   # (one shared empty CPU tensor, as upstream's torch.Tensor([]) per call)
        if shs is None:
            shs = empty
        if colors_precomp is None:
            colors_precomp = empty
        if scales is None:
            scales = empty
        if rotations is None:
            rotations = empty
        if cov3D_precomp is None:
            cov3D_precomp = empty
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, raster_settings, clamp_output, second, with_visibility)


# ---------------------------------------------------------------------------------------------
# introspection used by the stage-level parity tests (not part of the reference API)
def _debug_forward_state(means3D, opacities, raster_settings, shs=None, colors_precomp=None, scales=None,
                         rotations=None, cov3D_precomp=None):
    """Run forward and return (color, radii, dict of raw scratch sub-arrays as torch tensors)."""
    lib = _load()
    rs = raster_settings
    e = torch.empty(0, device=means3D.device)
    means2D = torch.zeros_like(means3D, requires_grad=True)
    holder = {}
    m3 = means3D.detach().requires_grad_(True)
    color, radii = _RasterizeGaussians.apply(m3, means2D, shs if shs is not None else e,
                                             colors_precomp if colors_precomp is not None else e, opacities,
                                             scales if scales is not None else e,
                                             rotations if rotations is not None else e,
                                             cov3D_precomp if cov3D_precomp is not None else e, rs)
    fn = color.grad_fn
    P, H, W = means3D.shape[0], int(rs.image_height), int(rs.image_width)
    geom = binning = image = None
    if P > 0:
        buf, (go, gl, io, il, bo, bl) = fn.scratch
        geom, image = buf[go:go + gl], buf[io:io + il]
        binning = buf[bo:bo + bl] if bl and 1 not in fn.bufs else fn.bufs[1]   # a too-small guess falls back to the callback
    N = color.grad_fn.num_rendered
    cap = color.grad_fn.binning_capacity if P > 0 else 0   # the binning buffer is laid out for `cap` >= N entries
    off = lambda name: lib.hgs_scratch_offset(name.encode(), P, cap, H, W)
    T = ((H + 15) // 16) * ((W + 15) // 16)

    def sub(buf, name, nbytes, dtype):
        o = off(name)
        return buf[o:o + nbytes].view(dtype)

    if P > 0:
        holder["splats"] = sub(geom, "splats", 64 * P, torch.float32).view(P, 16).contiguous()   # 64-byte records (hgs_common.h)
        holder["tiles_touched"] = sub(geom, "tiles_touched", 4 * P, torch.int32)
        holder["final_T"] = sub(image, "final_T", 4 * H * W, torch.float32).view(H, W)
        holder["n_contrib"] = sub(image, "n_contrib", 4 * H * W, torch.int32).view(H, W) & 0x0FFFFFFF  # top bits: clamp mask
        holder["ranges"] = sub(image, "ranges", 8 * T, torch.int32).view(T, 2)
        holder["seg_first"] = sub(image, "seg_first", 4 * (T + 1), torch.int32)   # checkpoint slots (meaningful when the frame left any)
        holder["has_checkpoints"] = bool(getattr(fn, "bw", None) is not None and fn.bw.state.ckpt)
        # the scan's decisions: [0] N, [2] long lists, [3] which tiles leave checkpoints (0 deep ones / 1 all / 2 none), [4] the long-list
        # threshold, [8] long tiles blended split by depth
        holder["n_total"] = sub(image, "n_total", 64, torch.int32)
        holder["sparse_frame"], holder["has_long_tiles"] = bool(fn.bw.state.sparse_frame), bool(fn.bw.state.has_long_tiles)
        holder["ckpt_slots_used"] = int(fn.bw.state.ckpt_slots_used)
        lst = sub(binning, "list", 8 * N, torch.int64)                      # sorted: (pos1 << 32) | mask << 28 | gaussian
        raw = lst & 0xFFFFFFFF
        holder["values"] = (raw & 0x0FFFFFFF).to(torch.int32)               # sorted list: Gaussian index of entry i
        holder["quad_masks"] = ((raw >> 28) & 0xF).to(torch.int32)          # conservative 8x8-quad coverage mask
        holder["pos1"] = (lst >> 32).to(torch.int32)                        # 1-based position inside the tile
        # the 64-bit key of entry i (tile << 32 | fp32 depth bits), re-assembled from what the device keeps:
        # the tile is the one whose range holds i, the depth is the Gaussian's
        counts = (holder["ranges"][:, 1] - holder["ranges"][:, 0]).long()
        tile_of = torch.repeat_interleave(torch.arange(T, device=lst.device), counts)
        depth_bits = holder["splats"][:, 9].contiguous().view(torch.int32)[holder["values"].long()].long() & 0xFFFFFFFF
        holder["keys"] = (tile_of << 32) | depth_bits
    holder["N"] = N
    holder["binning_capacity"] = cap
    return color.detach(), radii, holder
