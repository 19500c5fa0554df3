"""What every side-workload timing (tools/bench_c3.py, bench_c4.py) reports next to its milliseconds, the way bench.py does for the
headline workload: the frame's exact integers (N, visible), host busy time per step, the box, and a `roofline` block -- the dominant
stage's algorithmic bytes over its HIP-event time against the HBM peak."""
import os
import time
import zlib

import torch

import diff_gaussian_rasterization as dgr

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md
KERNEL_OF = {"sort": "tile_sort_small_kernel", "preprocess": "preprocess_kernel", "preprocess_backward": "preprocess_backward_kernel",
             "scan": "tile_scan_kernel", "emit_keys": "emit_scan_kernel", "blend_forward": "blend_forward_kernel"}


def kernel_of(stage, sparse=None, with_checkpoints=None):
    """the kernel of a stage as the rocprofv3 summaries name it; the blend backward's depends on the frame (last frame of this thread when not given)"""
    if stage != "blend_backward":
        return KERNEL_OF.get(stage)
    if sparse is None:
        cpp = dgr._load_cpp()
        _n, _cap, _long, sparse = last_frame()
        with_checkpoints = bool(cpp is not None and cpp.last_ckpt_info()[0] > 0)
    return backward_kernel(bool(sparse), bool(with_checkpoints))


def backward_kernel(sparse, with_checkpoints):
    """the kernel of the blend-backward stage by the frame's kind (blend.hip, launch_blend_backward): the name the rocprofv3 summaries carry"""
    if with_checkpoints:
        return "blend_backward_segmented_kernel" if sparse else "blend_backward_mixed_kernel"
    return "blend_backward_kernel"   # (<1>: one wave per quad on a sparse frame, <4>: one wave per tile)


def last_frame():
    """(N, capacity, has_long, sparse) of this thread's last forward, whichever binding ran it"""
    cpp = dgr._load_cpp()
    if cpp is not None:
        n, cap, has_long, sparse = cpp.last_frame_info()
        if n >= 0:
            return int(n), int(cap), bool(has_long), bool(sparse)
    n, cap = dgr.last_frame_info()
    return (int(n) if n is not None else None), (int(cap) if cap is not None else None), None, None


def _stat(name):
    return dgr._load().hgs_debug_stat(name.encode())


def timed_loop(step, steps):
    """-> (ms per step, host busy us per step): busy = wall time minus what the library spent waiting for N (the host's only idle
    time inside the loop); the GPU is drained before and after."""
    torch.cuda.synchronize()
    w0 = _stat("forward_wait_ns")
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    busy = (t_host * 1e9 - (_stat("forward_wait_ns") - w0)) / steps * 1e-3
    return ms, busy


def stage_bytes(P, Pv, N, S, T, K, M=16):
    """algorithmic bytes per launch of each stage (the fused sort + forward blend as in bench.py)"""
    return {"preprocess": P * (44 + 12 * K) + 8 * P + 67 * Pv, "preprocess_backward": Pv * (111 + 12 * K) + P * (40 + 12 * M),
            "blend_backward": 8 * T + 40 * N + 20 * S + 36 * Pv, "sort": 8 * T + 40 * N + 20 * S + 28 * N,
            "emit_keys": 36 * Pv + 8 * N, "scan": 8 * T}


def frame_bytes(P, Pv, N, S, T, K, M=16):
    return P * (108 + 12 * K + 12 * M) + Pv * (226 + 12 * K) + 124 * N + 40 * S + 24 * T


def roofline(stages_ms, P, Pv, N, H, W, D):
    """the `roofline` block of the stage that takes longest"""
    K, S, T = (D + 1) ** 2, H * W, ((H + 15) // 16) * ((W + 15) // 16)
    b = stage_bytes(P, Pv, N, S, T, K)
    dom = max((k for k in stages_ms if k in b), key=lambda k: stages_ms[k])
    gbps = b[dom] / (stages_ms[dom] * 1e-3) / 1e9
    kernel = kernel_of(dom)
    return {"bound": "hbm", "kernel": kernel, "stage": dom, "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(gbps / HBM_PEAK_GBPS, 5), "algorithmic_bytes_per_launch": int(b[dom]), "avg_launch_ms": stages_ms[dom],
            "traffic": None,   # (profiles/collect_workload.sh fills it in from its FETCH_SIZE / WRITE_SIZE passes of this workload)
            "note": "HIP events around every stage in a separate pass (each pair adds a few microseconds); the blend kernels are VALU / "
                    "latency bound, not HBM bound (DESIGN.md section 4)"}


_gpu_id = []


def gpu_unique_id():
    """'5ac0998dea87fceb' -- the visible GPU's unique id, or None.  From the environment when the caller provides it (HGS_GPU_UNIQUE_ID: the
    collection scripts ask `rocm-smi --showuniqueid` in the shell, before any GPU process exists); else from rocm-smi as a child process -- but ONLY
    while this process has not touched the GPU: rocm-smi is a python script, starting it is an exec, and an exec out of a process that has initialised
    the GPU runtime (torch.cuda.*, or rocprofv3's preloaded library) is what this pool's boxes refuse (round 6's first collection: ten refusals)."""
    if not _gpu_id:
        uid = os.environ.get("HGS_GPU_UNIQUE_ID") or None
        under_profiler = "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ)
        if uid is None and not torch.cuda.is_initialized() and not under_profiler:
            try:
                import re
                import subprocess
                out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showuniqueid"], capture_output=True, text=True, timeout=20).stdout
                m = re.search(r"Unique ID:\s*0x([0-9a-fA-F]+)", out)
                uid = m.group(1).lower() if m else None
            except Exception:
                pass
        _gpu_id.append(uid.lower().removeprefix("0x") if uid else None)
    return _gpu_id[0]


def box():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else round(int(q) / int(per), 1)
    except (OSError, ValueError):
        pass
    # which machine: the GPU's unique id (= its ASIC serial) and the host's boot id (the boxes share one container hostname).  The id comes from
    # `rocm-smi --showuniqueid` -- a child process, no GPU runtime in this one; the KFD topology in sysfs is not readable for the GPUs of the
    # host that this container was not given, and lists all of them (round 5 recorded null on every box)
    gpu_id, boot = gpu_unique_id(), None
    try:
        boot = open("/proc/sys/kernel/random/boot_id").read().strip()[:8]
    except OSError:
        pass
    return {"cpu_model": model, "host_cpus": os.cpu_count(), "cpu_quota": quota, "gpu_unique_id": gpu_id, "boot_id": boot, "box_id": "%04x" % (zlib.crc32(os.uname().nodename.encode()) & 0xFFFF),
            "frame_call": os.environ.get("HGS_FRAME_CALL", "1") != "0"}
