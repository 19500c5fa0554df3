"""Frame sharding across ranks (one process per GPU).

The reference renders frames one after another in a single process
(/root/reference/hugs/trainer/gs_trainer.py:463,551,616 -- validate / animate / render_canonical loops).
Frames are independent rasterizations of the same Gaussian set, so the multi-GPU form is: replicate
the Gaussians on every GPU, give frame i to rank i mod R, and exchange only per-frame scalars.  No
data-path collective exists; the gather below is the whole communication (RCCL on GPUs, gloo on CPU).
"""
import os

import torch
import torch.distributed as dist

# HGS_SHARDING_FORCE_COLLECTIVES=1: run the collectives even in a process group of ONE rank -- the only way to execute the RCCL
# code path (backend "nccl": broadcast, all_gather, all_reduce on device tensors) on a one-GPU box, where two ranks cannot
# share the device over RCCL (tests/test_gpu_configs.py, bench.py with HGS_BENCH_FORCE_PG=1)
_FORCE = os.environ.get("HGS_SHARDING_FORCE_COLLECTIVES") == "1"


def _single():
    return not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not _FORCE)


def frames_for_rank(num_frames, rank, world_size):
    """Round-robin: frame i belongs to rank i % world_size."""
    return list(range(rank, num_frames, world_size))


def broadcast_gaussians(tensors, src=0):
    """Replicate rank `src`'s Gaussian tensors on every rank (once, before the frame loop)."""
    if not _single():
        for t in tensors:
            dist.broadcast(t, src=src)
    return tensors


def gather_frame_metrics(frame_ids, values, num_frames, device=None):
    """All-gather per-frame scalars. `values` is [len(frame_ids), K]; returns [num_frames, K] on every rank
    with row i holding the metrics of frame i."""
    values = torch.as_tensor(values, dtype=torch.float64, device=device).reshape(len(frame_ids), -1)
    K = values.shape[1] if values.numel() else 0
    if _single():
        out = torch.zeros(num_frames, K, dtype=torch.float64, device=values.device)
        if len(frame_ids):
            out[torch.as_tensor(frame_ids, device=values.device)] = values
        return out
    world = dist.get_world_size()
    per_rank = (num_frames + world - 1) // world
    K = int(_agree_max(K, values.device))
    pad = torch.full((per_rank, K + 1), -1.0, dtype=torch.float64, device=values.device)
    if len(frame_ids):
        pad[:len(frame_ids), 0] = torch.as_tensor(frame_ids, dtype=torch.float64, device=values.device)
        pad[:len(frame_ids), 1:] = values
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    out = torch.zeros(num_frames, K, dtype=torch.float64, device=values.device)
    for b in bufs:
        ok = b[:, 0] >= 0
        out[b[ok, 0].long()] = b[ok, 1:]
    return out


def _agree_max(v, device):
    t = torch.tensor([float(v)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def max_over_ranks(seconds, device=None):
    """The slowest rank's time (the job's wall time)."""
    if _single():
        return float(seconds)
    return _agree_max(seconds, device if device is not None else torch.device("cpu"))
