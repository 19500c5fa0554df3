#!/usr/bin/env python3
"""Headline benchmark: rasterizer forward+backward FPS at 1080p, and the HBM roofline of its dominant kernel.

    python bench.py --gpus N --steps K --warmup W
    N > 1 either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...:
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment) or on its own: with WORLD_SIZE unset the parent
    process -- before it touches the GPU -- starts N child ranks itself, relays rank 0's JSON line and exits with the
    worst child's code.

Workload = BASELINE.json configs[1]: 200k scene Gaussians, [P,16,3] SH at degree 3, 1920x1080, white
background, seeded synthetic inputs already resident in HBM.  One step = one call of the
drop-in GaussianRasterizer (forward) + autograd backward with a fixed dL/dcolor, i.e. the work the
reference does at gs_renderer.py:144-152 and again inside loss.backward() (gs_trainer.py:287).
With N GPUs every rank renders its own camera of the same Gaussian set (frames are independent: weak
scaling); the only communication is the metric gather.  Rank 0 prints ONE JSON line on stdout.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ml-hugs_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def log(*a):
    print(*a, file=sys.stderr, flush=True)


KERNEL_OF = {"blend_backward": "blend_backward_kernel", "sort": "tile_sort_small_kernel"}   # stage -> kernel name in the profiles


def alg_bytes(P, Pv, N, S, T, K, M=16):
    """Algorithmic bytes: whole fwd+bwd frame, and the blend-backward kernel alone."""
    frame = P * (108 + 12 * K + 12 * M) + Pv * (226 + 12 * K) + 124 * N + 40 * S + 24 * T
    blend_bwd = 8 * T + 40 * N + 20 * S + 36 * Pv
    blend_fwd = 8 * T + 40 * N + 20 * S
    return frame, blend_bwd, blend_fwd


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (HGS_BENCH_STEPS: short runs under the PMC passes of profiles/collect_workload.sh)
    ap.add_argument("--steps", type=int, default=int(os.environ.get("HGS_BENCH_STEPS", 1000)))
    ap.add_argument("--warmup", type=int, default=3 if os.environ.get("HGS_BENCH_STEPS") else 50)
    ap.add_argument("--gaussians", type=int, default=200_000)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--forward-only", action="store_true")
    ap.add_argument("--cluster", type=float, default=0.0, help="fraction of the Gaussians in a central blob (not the headline workload)")
    ap.add_argument("--profile", choices=("uniform", "trained"), default="uniform",
                    help="uniform: SURVEY.md 8d's synthetic scene (the headline workload); trained: a scene shaped like what HUGS renders "
                         "after some thousand steps -- surfaces, heavy-tailed sizes, post-reset opacities, a 110 210-Gaussian human in "
                         "front (hugs_amd.synthetic.trained_scene_gaussians; --gaussians = the SCENE's count, SH degree 0 as the joint "
                         "render uses) -- not the headline workload")
    ap.add_argument("--spatial-order", action="store_true", help="store the Gaussians in 3-D Morton order (not the headline workload)")
    ap.add_argument("--frames-per-rank", type=int, default=1,
                    help="cameras each rank renders per step (frame f of rank r = camera r * K + f).  BASELINE configs[4] -- 8 frames x "
                         "300k Gaussians over 8 GPUs -- is `--gpus 8 --gaussians 300000 --frames-per-rank 1`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-two-streams", action="store_true", help="skip the secondary two-frames-in-flight figure")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget for the CPU baseline sample")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="ranks only rendezvous (gloo) and gather their ids -- no rasterizer work, no GPU needed; "
                         "checks the self-launcher (tests/test_bench_launcher.py)")
    return ap.parse_args()


def visible_gpu_count(sysfs="/sys/class/kfd/kfd/topology/nodes", devdir="/dev/dri", env=None):
    """GPUs this process could open, WITHOUT touching the GPU runtime (no torch, no HIP: a launcher parent that has
    initialised the GPU and then starts children is what this pool refuses).  KFD topology nodes with SIMDs are GPUs
    (CPU nodes have simd_count 0); one counts when its render node can be opened; *_VISIBLE_DEVICES lists cap the
    count.  None = cannot tell (no KFD topology readable): the caller then relies on the rank watcher."""
    env = os.environ if env is None else env
    try:
        nodes = sorted(os.listdir(sysfs))
    except OSError:
        return None
    count = 0
    for node in nodes:
        try:
            props = dict(line.split(None, 1) for line in open(os.path.join(sysfs, node, "properties")).read().splitlines() if " " in line)
        except OSError:
            continue
        if int(props.get("simd_count", "0").strip() or 0) <= 0:
            continue
        minor = int(props.get("drm_render_minor", "-1").strip() or -1)
        if minor >= 0 and os.path.isdir(devdir) and not os.access(os.path.join(devdir, f"renderD{minor}"), os.R_OK | os.W_OK):
            continue   # a GPU of the host that this container was not given
        count += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(var)
        if v is not None:
            count = min(count, len([x for x in v.split(",") if x.strip() != ""]))
    return count


def launch_ranks(args):
    """--gpus N without a launcher: start N fresh rank processes (never exec / re-exec), one per LOCAL_RANK, rendezvous on
    127.0.0.1.  Rank 0's stdout is this process's stdout.  The parent never touches the GPU runtime and never imports torch:
    GPUs are counted from the KFD topology in sysfs (visible_gpu_count)."""
    import socket
    import subprocess
    n = args.gpus
    backend = os.environ.get("HGS_BENCH_BACKEND", "nccl")
    if backend == "nccl" and not args.launcher_selftest:
        have = visible_gpu_count()
        if have is None and not os.path.exists("/dev/kfd"):
            have = 0   # no KFD device at all: there is no GPU to give a rank
        if have is not None and have < n:
            raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible. One rank per GPU over RCCL needs {n}; "
                             "set HGS_BENCH_BACKEND=gloo to let ranks share GPUs (plumbing check, not a scaling number).")
        if have is None:
            log("[launcher] could not count GPUs from the KFD topology: relying on the rank watcher")
    assert "torch" not in sys.modules, "the launcher parent must stay free of torch / the GPU runtime"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    log(f"[launcher] started {n} ranks (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}, backend {backend}; "
        f"parent imported torch: {'torch' in sys.modules}")
    # Watch ALL ranks (what torchrun does): the first one that fails takes the others down with it -- a rank that died in
    # init would otherwise leave the rest in a collective until the process-group timeout -- and there is an overall deadline.
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + float(os.environ.get("HGS_BENCH_DEADLINE_S", "3600"))
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = f"rank {r} exited with code {p.returncode}"
        if failed is None and time.monotonic() > deadline:
            failed = "deadline (HGS_BENCH_DEADLINE_S) passed"
        time.sleep(0.05)
    if failed is None:
        bad = [r for r, p in enumerate(procs) if p.returncode != 0]
        failed = f"rank {bad[0]} exited with code {procs[bad[0]].returncode}" if bad else None
    if failed is not None:
        log(f"[launcher] {failed}: stopping the other ranks")
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=10.0)
    codes = [p.returncode for p in procs]
    for line in (out0[0] if out0 else b"").decode().splitlines():   # the contract is ONE JSON line on stdout: library chatter goes to stderr
        if line.startswith("{") and failed is None:
            print(line, flush=True)
        elif line.strip():
            log(line)
    if failed is not None or any(codes):
        log(f"[launcher] rank exit codes {codes}")
        raise SystemExit(max([abs(c) for c in codes if c] + [1]))


def launcher_selftest(args, rank, world):
    """Rendezvous + the bench's own gather plumbing on gloo, nothing else."""
    import torch
    import torch.distributed as dist
    from hugs_amd import sharding
    if os.environ.get("HGS_SELFTEST_FAIL_RANK") == str(rank):   # tests/test_bench_launcher.py: a rank that dies before the rendezvous
        raise SystemExit(7)
    if os.environ.get("HGS_SELFTEST_HANG_RANK") == str(rank):   # ... and one that never comes back (the launcher's deadline)
        time.sleep(3600)
    if world > 1:
        import datetime
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=float(os.environ.get("HGS_BENCH_INIT_TIMEOUT_S", "120"))))
    ids = sharding.gather_frame_metrics([rank], [[float(rank), float(os.getpid())]], world, device=torch.device("cpu"))
    slowest = sharding.max_over_ranks(0.001 * (rank + 1), torch.device("cpu"))
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "ranks_seen": int((ids[:, 1] > 0).sum()),
                          "rank_ids": [int(x) for x in ids[:, 0].tolist()], "max_over_ranks_ok": abs(slowest - 0.001 * world) < 1e-9}),
              flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def host_cpu():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count() or 1


_box = []


def box_of_this_run():
    """which machine: CPU model, quota, the GPU's unique id (rocm-smi, a child process), the host's boot id (tools/bench_common.py)"""
    if _box:
        return _box[0]
    _box.append(_box_of_this_run())
    return _box[0]


def _box_of_this_run():
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_common
        return bench_common.box()
    except Exception as e:   # (never the reason a bench line is lost)
        return {"error": str(e)[:200]}


def cgroup_cpu_stat(path="/sys/fs/cgroup/cpu.stat"):
    """{nr_periods, nr_throttled, throttled_usec, usage_usec, ...} of the cgroup this process runs in (cgroup v2); {} if unreadable"""
    try:
        return {k: int(v) for k, v in (line.split() for line in open(path).read().splitlines() if len(line.split()) == 2)}
    except (OSError, ValueError):
        return {}


def thread_cpu_seconds():
    """{tid: (name, user + system CPU seconds)} of this process's threads (/proc/self/task)"""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                raw = open(f"/proc/self/task/{tid}/stat").read()
                name = raw[raw.index("(") + 1:raw.rindex(")")]
                f = raw[raw.rindex(")") + 2:].split()
                out[int(tid)] = (name, (int(f[11]) + int(f[12])) / tick)
            except (OSError, ValueError, IndexError):
                pass
    except OSError:
        pass
    return out


def cgroup_cpu_max(path="/sys/fs/cgroup/cpu.max"):
    try:
        q, per = open(path).read().split()
        return None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        return None


def measured_copy_peak(torch, device, nbytes=1 << 30, reps=10):
    """Device-to-device copy rates (read + write bytes / time) of this GPU, now -- the practical HBM ceiling next to the
    datasheet figure: (the library's float4-per-thread copy kernel -- the shape MI355X_MICROARCH.md measures 6.29 TB/s
    with --, torch's Tensor.copy_)."""
    import ctypes
    import diff_gaussian_rasterization as dgr
    lib = dgr._load()
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=device).normal_()
    b = torch.empty_like(a)
    stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)   # the events below are recorded on the same stream

    def rate(copy):
        for _ in range(3):
            copy()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            copy()
        e1.record()
        e1.synchronize()
        return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9

    def lib_copy():
        if lib.hgs_copy_bandwidth(b.data_ptr(), a.data_ptr(), nbytes, stream) < 0:
            raise RuntimeError(lib.hgs_last_error().decode())

    return rate(lib_copy), rate(lambda: b.copy_(a))


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.launcher_selftest:
        return launcher_selftest(args, rank, world)
    # HGS_BENCH_CPUS_TOTAL=n: every rank's affinity is cut to the SAME n CPUs (the first n this process may use) before torch or the GPU
    # runtime exist in it -- no wrapper process, no exec: how "8 ranks on half the box's CPU quota" is measured
    cpus_total = int(os.environ.get("HGS_BENCH_CPUS_TOTAL", "0"))
    if cpus_total > 0:
        assert "torch" not in sys.modules
        os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:cpus_total])

    import torch
    import torch.distributed as dist

    if rank == 0:
        box_of_this_run()   # (asks rocm-smi for the GPU's id NOW: a child process, which must not be started once this process has touched the GPU)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # HGS_BENCH_BACKEND=gloo lets the multi-rank code path be exercised on a box with fewer GPUs than ranks (several
    # ranks then share a GPU); the driver's runs use the default: one rank per GPU over RCCL ("nccl")
    backend = os.environ.get("HGS_BENCH_BACKEND", "nccl")
    device = torch.device("cuda", local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank)
    torch.cuda.set_device(device)
    # HGS_BENCH_FORCE_PG=1 (with HGS_SHARDING_FORCE_COLLECTIVES=1): a process group of ONE rank, collectives and all -- how the
    # RCCL branch of this file gets executed on a one-GPU lease (tests/test_gpu_configs.py)
    force_pg = world == 1 and os.environ.get("HGS_BENCH_FORCE_PG") == "1"
    if force_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_pg:
        import datetime
        tmo = datetime.timedelta(seconds=float(os.environ.get("HGS_BENCH_INIT_TIMEOUT_S", "300")))  # a dead sibling must not hang the rest
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
    log(f"[rank {rank}] world_size {dist.get_world_size() if (world > 1 or force_pg) else 1} ({backend if (world > 1 or force_pg) else 'single process'}), "
        f"device {device} = {torch.cuda.get_device_name(device)}")

    from diff_gaussian_rasterization import (GaussianRasterizationSettings, GaussianRasterizer, profile_enable,
                                             profile_read)
    from hugs_amd import sharding, synthetic as syn

    P, H, W, D = args.gaussians, args.height, args.width, args.sh_degree
    cam0 = syn.pinhole_camera(H, W)
    if args.profile == "trained":
        g = syn.trained_scene_gaussians(P, cam0, seed=0)
        P = g["means3D"].shape[0]   # scene + human
        D = 0                       # (render_human_scene takes the HUMAN model's active degree: 0 in the release configs)
    else:
        g = syn.scene_gaussians(P, cam0, seed=0, sigma_px=4.0, cluster=args.cluster)
    if args.spatial_order:   # not the headline workload: the same Gaussians stored in 3-D Morton order
        from hugs_amd.spatial import morton_order
        order = morton_order(g["means3D"])
        g = {k: (v[order] if isinstance(v, np.ndarray) and v.shape[:1] == (P,) else v) for k, v in g.items()}
    # frame k of the batch: the same scene seen from a slightly yawed camera (frame 0 = identity pose); rank r renders
    # frames r * K .. r * K + K - 1 every step (hugs_amd.sharding: frames are independent, no data-path collective)
    KF = max(1, args.frames_per_rank)

    def camera_of(frame):
        yaw = math.radians(1.5) * frame
        w2c = np.eye(4)
        w2c[0, 0], w2c[0, 2], w2c[2, 0], w2c[2, 2] = math.cos(yaw), math.sin(yaw), -math.sin(yaw), math.cos(yaw)
        return syn.camera_from_w2c(w2c, cam0["fovx"], cam0["fovy"], H, W)

    cams = [camera_of(rank * KF + f) for f in range(KF)]
    cam = cams[0]
    dL = syn.pixel_grad(H, W)

    dev = lambda a, grad=False: torch.from_numpy(np.ascontiguousarray(a)).to(device).requires_grad_(grad)
    t = {k: dev(g[k], True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    if backend == "nccl" and (world > 1 or force_pg):
        sharding.broadcast_gaussians([v.data for v in t.values()])  # replicas of rank 0's Gaussians (same seed anyway)
    means2D = torch.zeros(P, 3, device=device, requires_grad=True)
    dLd = dev(dL)
    bg_white = torch.ones(3, device=device)
    all_settings = [GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(c["fovx"] * 0.5), tanfovy=math.tan(c["fovy"] * 0.5),
        bg=bg_white, scale_modifier=1.0, viewmatrix=dev(c["world_view_transform"]),
        projmatrix=dev(c["full_proj_transform"]), sh_degree=D, campos=dev(c["camera_center"]),
        prefiltered=False, debug=False) for c in cams]
    settings = all_settings[0]
    leaves = list(t.values()) + [means2D]

    def step():
        for st_ in all_settings:   # this rank's K frames
            rast = GaussianRasterizer(raster_settings=st_)  # a new module per call, as the reference does
            color, radii = rast(means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], shs=t["shs"],
                                scales=t["scales"], rotations=t["rotations"])
            if not args.forward_only:
                color.backward(dLd)
                for x in leaves:
                    x.grad = None
        return color, radii

    def fence():
        if world > 1 or force_pg:
            dist.barrier()
        torch.cuda.synchronize()

    # Before the W warmup steps the contract asks for: bring the GPU to its sustained clocks.  A run of 20 steps straight after
    # start-up measured 1 950-1 980 FPS where 1 000 steps of the same process on the same box measure 2 090: the kernels themselves
    # take 4 % longer at first (the library's HIP events around the blend backward: 0.2485 against 0.238 ms).  Untimed, reported in the line.
    prewarm_s = float(os.environ.get("HGS_BENCH_PREWARM_S", "0.25"))
    prewarm_frames, t_pre = 0, time.perf_counter()
    while prewarm_s > 0 and (prewarm_frames < 64 or time.perf_counter() - t_pre < prewarm_s):
        step()
        prewarm_frames += KF
        if (prewarm_frames // KF) % 16 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    prewarm_took = time.perf_counter() - t_pre
    for _ in range(args.warmup):
        step()
    # (forward-only: the forward blend runs fused with the tile sort, stage "sort" = tile_sort_small_kernel)
    dominant = "sort" if args.forward_only else "blend_backward"
    # live HIP-event timing of the dominant kernel over the timed region: on every 8th launch of a long run (an event pair
    # costs ~5 us of GPU time around the kernel it brackets: timing every launch took 2 % off the throughput it was measured
    # beside), more often on a short one (at least 8 launches are timed whenever --steps >= 8)
    every_nth = max(1, min(8, args.steps * KF // 8))   # (20 steps: every 2nd launch, ten samples)
    profile_enable((dominant,), every_nth=every_nth)
    import diff_gaussian_rasterization as dgr
    _lib = dgr._load()
    torch.cuda.reset_peak_memory_stats(device)
    fence()
    wait0 = _lib.hgs_debug_stat(b"forward_wait_ns")
    cpu0, cg0 = time.process_time(), cgroup_cpu_stat()   # (process-wide user + system CPU seconds: every thread of this rank)
    thr0 = thread_cpu_seconds()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        color, radii = step()
    host_issue_s = time.perf_counter() - t0   # (the loop without the final drain)
    fence()
    elapsed = time.perf_counter() - t0
    cpu_seconds, cg1 = time.process_time() - cpu0, cgroup_cpu_stat()
    thr1 = thread_cpu_seconds()
    # which of rank 0's threads the CPU time went to (the main thread issues frames and waits for N; "pt_autograd_*" runs the backward node)
    threads_cpu = sorted(((n, round(t - thr0.get(tid, (n, 0.0))[1], 3)) for tid, (n, t) in thr1.items()), key=lambda x: -x[1])
    threads_cpu = [x for x in threads_cpu if x[1] >= 0.01][:8]
    # host busy per frame: the loop's wall time minus what the library spent waiting for N (the host's only idle time inside it)
    host_busy_us = (host_issue_s * 1e9 - (_lib.hgs_debug_stat(b"forward_wait_ns") - wait0)) / (args.steps * KF) * 1e-3
    peak_mem = torch.cuda.max_memory_allocated(device)
    _cpp = dgr._load_cpp()
    ckpt_bytes, ckpt_used = _cpp.last_ckpt_info() if _cpp is not None else (None, None)
    prof = profile_read()
    profile_enable(())
    coll_dev = device if backend == "nccl" else torch.device("cpu")  # where the tiny metric collectives run
    own_elapsed = elapsed   # this rank's own clock over the bracket (a straggler shows in per_rank_fps)
    elapsed = sharding.max_over_ranks(elapsed, coll_dev)

    # exact integers of this frame (shared with the oracle): N and the visible count
    N = dgr.last_frame_info()[0]
    Pv = int((radii > 0).sum())
    if N is None:
        from diff_gaussian_rasterization import _debug_forward_state
        N = _debug_forward_state(t["means3D"].detach(), t["opacities"].detach(), settings, shs=t["shs"].detach(),
                                 scales=t["scales"].detach(), rotations=t["rotations"].detach())[2]["N"]
    frames = sharding.gather_frame_metrics([rank], [[float(N), float(Pv), float(device.index), 1.0, KF * args.steps / own_elapsed, host_busy_us,
                                                     cpu_seconds, float(len(os.sched_getaffinity(0)))]],
                                           world, device=coll_dev)
    copy_peak, torch_copy_peak = measured_copy_peak(torch, device) if rank == 0 else (None, None)

    # per-stage breakdown in a separate, untimed pass (every stage bracketed by events)
    profile_enable()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    stages = {k: round(v[0] / 10.0, 4) for k, v in profile_read().items()}  # ms per frame (a stage may have >1 timed span)
    profile_enable(())

    # secondary figure (single process only): the same frames with two in flight on two HIP streams -- independent frames
    # (validation / animation loops, or the reference's separate human + scene renders) can overlap one frame's
    # latency-bound binning with another's VALU-bound blending.  `value` above stays the serial number.
    fps_two_streams = None
    if world == 1 and not args.forward_only and not args.no_two_streams:
        # Every stream renders its OWN copy of the leaves, made on that stream: autograd creates a leaf's AccumulateGrad node on the stream
        # that is current when the leaf first enters a graph, and warns when a backward then runs on another one (round 5's leg shared the
        # default stream's leaves between the two side streams: the warning was in every driver run).
        side = [torch.cuda.Stream(device) for _ in range(2)]
        side_leaves = []
        for st_ in side:
            st_.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(st_):
                tt = {k: v.detach().clone().requires_grad_(True) for k, v in t.items()}
                side_leaves.append((tt, torch.zeros(P, 3, device=device, requires_grad=True)))

        def step_on(k):
            tt, m2 = side_leaves[k]
            with torch.cuda.stream(side[k]):
                for st2 in all_settings:
                    color2, _ = GaussianRasterizer(raster_settings=st2)(means3D=tt["means3D"], means2D=m2, opacities=tt["opacities"], shs=tt["shs"],
                                                                      scales=tt["scales"], rotations=tt["rotations"])
                    color2.backward(dLd)
                    for x in list(tt.values()) + [m2]:
                        x.grad = None

        for st_ in side:
            st_.wait_stream(torch.cuda.current_stream(device))   # (dLd and the settings' tensors were made on the caller's stream)
        for k in range(20):
            step_on(k & 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step_on(k & 1)
        torch.cuda.synchronize()
        fps_two_streams = args.steps / (time.perf_counter() - t0)

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    K = (D + 1) ** 2
    S, T = H * W, ((H + 15) // 16) * ((W + 15) // 16)
    frame_B, bwd_B, fwd_B = alg_bytes(P, Pv, N, S, T, K)
    dom_ms = prof[dominant][0] / prof[dominant][1]
    # the blend backward's kernel depends on the frame (a dense frame with deep tiles -- the trained profile -- runs the one-launch mixed
    # kernel, a sparse one the segmented kernel): the name the rocprofv3 summaries of this workload carry, from this thread's last frame
    KERNEL_OF = dict(globals()["KERNEL_OF"])
    if dominant == "blend_backward":
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_common
            KERNEL_OF["blend_backward"] = bench_common.kernel_of("blend_backward")
        except Exception:
            pass
    dom_B = fwd_B + 28 * N if args.forward_only else bwd_B   # fused tile sort + forward blend: + keys in, list and compacted lists out
    achieved = dom_B / (dom_ms * 1e-3) / 1e9
    fps = world * KF * args.steps / elapsed
    out = {
        "metric": "rasterizer fwd+bwd FPS @1080p vs #Gaussians; achieved HBM GB/s vs peak",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "prewarm": {"frames": prewarm_frames, "seconds": round(prewarm_took, 3),
                    "note": "untimed frames in front of the W warmup steps (GPU clock ramp); HGS_BENCH_PREWARM_S=0 turns it off"},
        "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "ranks_seen": int(frames[:, 3].sum()), "expected_ranks": args.gpus,
        "backend": backend if world > 1 else (backend + " (process group of one rank)" if force_pg else "single process"),
        "rank_devices": [int(x) for x in frames[:, 2].tolist()],
        "per_rank_N": [int(x) for x in frames[:, 0].tolist()],
        "per_rank_fps": [round(float(x), 2) for x in frames[:, 4].tolist()],   # each rank's own clock: a straggler is visible here
        "config": {"workload": f"{'trained-scene profile (not a BASELINE config)' if args.profile == 'trained' else 'configs[1]' if (P, KF) == (200_000, 1) else ('configs[4] (frame batch)' if P == 300_000 else 'sweep point')}: "
                               f"{P} scene Gaussians, {W}x{H}, SH degree {D} on [P,16,3], "
                               f"{'forward only' if args.forward_only else 'forward+backward'} through "
                               f"GaussianRasterizer (drop-in API), {KF} camera(s) per GPU and step",
                   "frames_per_rank": KF, "frames_per_step_all_ranks": world * KF,
                   "gaussians": P, "visible": Pv, "num_rendered_N": int(N), "tiles": T,
                   "N_per_frame_all_ranks": [int(x) for x in frames[:, 0].tolist()]},
        "roofline": {"bound": "hbm", "kernel": KERNEL_OF[dominant], "achieved": round(achieved, 2),
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                     "peak_measured": round(copy_peak, 1), "frac_of_measured": round(achieved / copy_peak, 5),
                     "peak_measured_how": "1 GiB device-to-device float4-per-thread copy kernel (hgs_copy_bandwidth) on this GPU in "
                                          "this run, (read + write bytes) / time; peak_torch_copy = the same with Tensor.copy_",
                     "peak_torch_copy": round(torch_copy_peak, 1),
                     "traffic": None, "algorithmic_bytes_per_launch": int(dom_B),
                     "avg_launch_ms": round(dom_ms, 4), "launches_timed": prof[dominant][1], "timed_every_nth_launch": every_nth,
                     "avg_launch_note": "HIP events on the launch stream around the kernel, inside the timed region; the event pair "
                                        "itself adds ~1-2 % to the figure (rocprofv3's kernel trace in profiles/ is the event-free one)",
                     "note": "north_star's '>= 60 % of the HBM roofline' is structurally unreachable for the two blend kernels: "
                             "they do ~256 pixel-splat evaluations per 40-byte list entry and are VALU bound (see "
                             "roofline_valu: pipes saturated at ~0.47 of peak issue, at the floor of their instruction mix) with HBM mostly idle; the per-Gaussian kernels (K1, K8) are the ones on the "
                             "HBM roofline (DESIGN.md section 4)"},
        "whole_frame": {"algorithmic_bytes": int(frame_B), "GB_per_s": round(frame_B * fps / world / 1e9, 2),
                        "frac_of_hbm_peak": round(frame_B * fps / world / 1e9 / HBM_PEAK_GBPS, 5),
                        "frac_of_measured_peak": round(frame_B * fps / world / 1e9 / copy_peak, 5)},
        "box": box_of_this_run(),
        "host_busy_us_per_frame": round(host_busy_us, 1),
        # the host side of N ranks on one box (every rank's own figures; rank 0 reads the cgroup the ranks share): CPU seconds are process-wide
        # user + system time over the timed region, `cores_busy` = their sum over the region's wall time -- against the cgroup's quota
        "ranks_host": {"host_busy_us_per_frame": [round(float(x), 1) for x in frames[:, 5].tolist()],
                       "cpu_seconds": [round(float(x), 4) for x in frames[:, 6].tolist()],
                       "cores_busy": round(float(frames[:, 6].sum()) / elapsed, 3), "wall_seconds": round(elapsed, 4),
                       "rank0_threads_cpu_seconds": threads_cpu,
                       "affinity_cpus_per_rank": [int(x) for x in frames[:, 7].tolist()], "cpus_total_requested": cpus_total or None,
                       "cgroup": {"cpu_max": cgroup_cpu_max(), "nr_periods": cg1.get("nr_periods", 0) - cg0.get("nr_periods", 0),
                                  "nr_throttled": cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0),
                                  "throttled_usec": cg1.get("throttled_usec", 0) - cg0.get("throttled_usec", 0)} if cg0 or cg1 else None},
        "memory": {"peak_allocated_MB": round(peak_mem / 2**20, 1), "checkpoint_buffer_MB": None if ckpt_bytes is None else round(ckpt_bytes / 2**20, 1),
                   "checkpoint_slots_used": ckpt_used},
        "pixel_splat_evals_per_s": round(256.0 * N * (1 if args.forward_only else 2) * fps / world, 1),
        "stages_ms": stages,
        # the two per-Gaussian kernels are the ones on the HBM roofline: their algorithmic bytes
        # (K1 = P (44 + 12 K) + 8 P + 67 P_v, K8 = P_v (111 + 12 K) + P (40 + 12 M)) over the live event timings
        "stage_rooflines": {name: {"algorithmic_bytes": int(b), "GB_per_s": round(b / (stages[name] * 1e-3) / 1e9, 1),
                                   "frac_of_hbm_peak": round(b / (stages[name] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                   "frac_of_measured": round(b / (stages[name] * 1e-3) / 1e9 / copy_peak, 4)}
                            for name, b in (("preprocess", P * (44 + 12 * K) + 8 * P + 67 * Pv),
                                            ("preprocess_backward", Pv * (111 + 12 * K) + P * (40 + 12 * 16)))
                            if stages.get(name) and not (args.forward_only and name == "preprocess_backward")},
        "stage_rooflines_note": "preprocess also zeroes the backward's [P,12] gradient accumulator (48 B per Gaussian, fused into K1 instead of a memset "
                                "launch): SURVEY 8d's K1 bytes do not count it -- `with_accumulator_zeroing` does (the WRITE_SIZE counter sees it: 1.9x the 8d writes)",
        "stages_ms_note": "separate untimed pass with an event pair around EVERY stage: each pair costs a few microseconds of GPU "
                          "time, so the sum exceeds ms_per_step",
    }
    if "preprocess" in out["stage_rooflines"] and not args.forward_only:
        b = P * (44 + 12 * K) + 8 * P + 67 * Pv + 48 * P
        gb = b / (stages["preprocess"] * 1e-3) / 1e9
        out["stage_rooflines"]["preprocess"]["with_accumulator_zeroing"] = {"bytes": int(b), "GB_per_s": round(gb, 1), "frac_of_hbm_peak": round(gb / HBM_PEAK_GBPS, 4),
                                                                            "frac_of_measured": round(gb / copy_peak, 4)}
    if fps_two_streams is not None:
        out["two_frames_in_flight"] = {"value": round(fps_two_streams, 2), "unit": "frames/s",
                                       "note": "same workload, frames alternate between two HIP streams of one process"}

    # HBM traffic and VALU issue utilisation of the dominant kernel from the committed rocprofv3 PMC passes (separate
    # runs of this same command; collected and corrected as MI355X_MICROARCH.md prescribes).  Attached only when they
    # were taken on this workload AND on the kernel sources of the running build (profiles/build_id.py).
    import glob
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    from build_id import csrc_sha16
    here = csrc_sha16()

    def newest(pattern, fits=None):
        """the newest file of the pattern (named per round: r1p < r2a < r2b) -- with `fits`, the newest whose contents are of THIS workload (a
        collection leaves one file per tracked workload: r6w_pmc_traffic.json, r6w_trained_pmc_traffic.json, ...), else the newest at all"""
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
        if fits:
            for cand in reversed(files):
                try:
                    if fits(json.load(open(cand))):
                        return cand
                except Exception:
                    pass
        return files[-1] if files else None

    def of_this_workload(doc):
        w = doc.get("workload") or {}
        return (w.get("gaussians"), w.get("height"), w.get("width"), w.get("sh_degree")) == (P, H, W, D) and args.profile == w.get("profile", "uniform")

    f = newest("*_pmc_traffic.json", of_this_workload)
    if f:
        pmc = json.load(open(f))
        wl = pmc.get("workload", {})
        same_wl = (wl.get("gaussians"), wl.get("height"), wl.get("width"), wl.get("sh_degree")) == (P, H, W, D)
        if same_wl and pmc.get("csrc_sha16") == here and KERNEL_OF[dominant] in pmc["kernels"]:
            out["roofline"]["traffic"] = pmc["kernels"][KERNEL_OF[dominant]]["hbm_bytes_corrected"]
            out["roofline"]["traffic_source"] = "profiles/" + os.path.basename(f)
        else:
            out["roofline"]["traffic_note"] = (f"stale or other workload: profiles/{os.path.basename(f)} was taken on csrc "
                                               f"{pmc.get('csrc_sha16')}, this build is {here}")
    # The bound that holds for the blend kernels is VALU, reported as a roofline of its own:
    #   achieved_frac  = wave-instructions (PMC) x 2 cycles (the guide's issue rate: one wave64 VALU instruction per SIMD every 2 cycles)
    #                    / (1 024 SIMDs x kernel cycles)
    #   mix_floor_frac = wave-instructions x the measured cost of THIS kernel's instruction mix (profiles/*_valu_mix.json: static histogram
    #                    of its hot loop x profiles/r2_valu_model.txt) / 1 024 SIMDs / kernel time: ~1 = the kernel sits on what its mix allows
    #   valu_pipe_busy_frac = SQ_ACTIVE_INST_VALU x 4 / (SIMDs x cycles): the pipes never idle -- NOT a fraction of peak issue
    # Attached only for the workload and the kernel sources the counters were taken on.
    f = newest("*_valu_utilization.json", of_this_workload)
    if f:
        vu_all = json.load(open(f))
        vu = vu_all.get("kernels", {}).get(KERNEL_OF[dominant])
        wl = vu_all.get("workload", {})
        same_wl = (wl.get("gaussians"), wl.get("height"), wl.get("width"), wl.get("sh_degree")) == (P, H, W, D) and args.profile == wl.get("profile", "uniform")
        if vu and vu_all.get("csrc_sha16") == here and same_wl:
            instr, cycles = float(vu["valu_wave_instructions"]), float(vu["kernel_cycles"])
            rv = {"bound": "valu", "kernel": KERNEL_OF[dominant], "wave_instructions_per_launch": instr, "kernel_cycles": cycles, "simds": 1024,
                  "issue_peak": "one wave64 VALU instruction per SIMD every 2 cycles (MI355X_MICROARCH.md)",
                  "achieved_frac": round(instr * 2.0 / (1024.0 * cycles), 4), "valu_pipe_busy_frac": vu["valu_busy_frac"],
                  "source": "profiles/" + os.path.basename(f) + " (PMC: SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, GRBM_GUI_ACTIVE; separate rocprofv3 passes)"}
            fm = newest("*_valu_mix.json")
            if fm:
                mix = json.load(open(fm))
                mk = mix.get("kernels", {}).get(KERNEL_OF[dominant])
                if mk and mix.get("csrc_sha16") == here:
                    rv["mix_ns_per_instruction"] = mk["mix_ns_per_instruction"]
                    rv["mix_floor_frac"] = round(instr * mk["mix_ns_per_instruction"] * 1e-9 / 1024.0 / (dom_ms * 1e-3), 4)
                    rv["mix_source"] = "profiles/" + os.path.basename(fm) + f" ({mk['scope']}: " + ", ".join(f"{k} {v['share']:.0%}" for k, v in mk["classes"].items()) + ")"
            out["roofline_valu"] = rv
        else:
            out["roofline_valu"] = {"bound": "valu", "kernel": KERNEL_OF[dominant], "achieved_frac": None,
                                    "note": f"profiles/{os.path.basename(f)} was taken on csrc {vu_all.get('csrc_sha16')} / workload {wl}; this build is {here}, "
                                            f"this run {P} Gaussians {W}x{H} degree {D} profile {args.profile}"}

    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(g, cam, dL, H, W, D, args.cpu_seconds, not args.forward_only)
    print(json.dumps(out), flush=True)
    if world > 1 or force_pg:
        dist.barrier()
        dist.destroy_process_group()
    if out["ranks_seen"] != args.gpus:   # a line that LOOKS like an N-GPU number but is not one must not exit 0
        log(f"bench.py: {out['ranks_seen']} rank(s) reported, --gpus {args.gpus} expected")
        raise SystemExit(3)


def cpu_baseline(g, cam, dL, H, W, D, budget_s, with_backward):
    """The oracle (a CPU port of the same algorithm -- the reference has no CPU path) timed on
    this box's host cores on a bounded sample of the same workload. Baseline only."""
    from oracle import hgs_oracle as ho
    # all cores this process may use (round 4: the port's pixel backward adds tile-local sums into ONE shared double accumulator --
    # the per-thread accumulators of round 3 capped it at 32 threads -- and its per-Gaussian loops, key emission and radix sort
    # run under OpenMP too).  `cores` = threads actually used = the affinity mask capped by the cgroup CPU quota: the GPU boxes
    # show 256 CPUs under a 16-CPU quota, where 256 threads take 1.70 s per frame and 16 take 0.49 s.
    cores = ho.usable_cpus()
    ho.set_threads(cores)
    inp = ho.Inputs(g["means3D"], g["opacities"], cam["world_view_transform"], cam["full_proj_transform"],
                    cam["camera_center"], math.tan(cam["fovx"] * 0.5), math.tan(cam["fovy"] * 0.5), H, W,
                    np.ones(3, np.float32), shs=g["shs"], scales=g["scales"], rotations=g["rotations"], sh_degree=D)
    n, t0 = 0, time.perf_counter()
    while True:
        f = ho.forward(inp)
        if with_backward:
            ho.backward(inp, f, dL)
        n += 1
        el = time.perf_counter() - t0
        if el + el / n > budget_s or n >= 8:
            break
    log(f"[cpu_baseline] {n} frame(s) in {el:.2f} s on {cores} threads")
    model, host_cpus = host_cpu()
    return {"value": round(n / el, 4), "unit": "frames/s", "cores": cores, "host_cpus": host_cpus, "cpu_quota": cores if cores < host_cpus else None, "cpu_model": model, "kind": "port",
            "sample": f"{n} full frame(s) of the same workload ({'fwd+bwd' if with_backward else 'fwd'}), "
                      "C oracle (oracle/hgs_oracle.c, fp32, OpenMP: per-Gaussian loops, chunked radix sort, blend kernels over tiles)"}


if __name__ == "__main__":
    main()
