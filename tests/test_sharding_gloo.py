"""CPU: the multi-rank path (frame sharding + metric gather) on the gloo backend, world_size 2.
On GPUs the same code runs over RCCL; there is no data-path collective to test beyond this."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "ml-hugs_amd"))
    from hugs_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        num_frames = 5
        mine = sharding.frames_for_rank(num_frames, rank, world)
        # replicas: rank 0's Gaussians reach every rank
        t = torch.full((4, 3), float(rank + 1))
        sharding.broadcast_gaussians([t], src=0)
        # per-frame metrics: (frame id squared, rank)
        vals = [[float(f * f), float(rank)] for f in mine]
        table = sharding.gather_frame_metrics(mine, vals, num_frames)
        slow = sharding.max_over_ranks(1.0 + rank)
        ret[rank] = (mine, t.clone(), table.clone(), slow)
    finally:
        dist.destroy_process_group()


def test_frame_sharding_and_metric_gather_world2():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0][0] == [0, 2, 4] and ret[1][0] == [1, 3]
    for r in range(world):
        mine, t, table, slow = ret[r]
        assert torch.all(t == 1.0)  # broadcast from rank 0
        assert slow == 2.0          # max over ranks
        expect = np.array([[f * f, f % world] for f in range(5)], np.float64)
        assert np.array_equal(table.numpy(), expect)


def test_single_process_paths_need_no_process_group():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "ml-hugs_amd"))
    from hugs_amd import sharding
    assert sharding.frames_for_rank(3, 0, 1) == [0, 1, 2]
    tab = sharding.gather_frame_metrics([0, 1, 2], [[1.0], [2.0], [3.0]], 3)
    assert tab.shape == (3, 1) and tab[2, 0] == 3.0
    assert sharding.max_over_ranks(0.5) == 0.5
