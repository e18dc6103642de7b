"""CPU, world_size 2 (gloo): clip sharding + per-round all_gather reassembles outputs in dataset order,
including the ragged tail (n_clips % world != 0)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from unigeo_amd.shard import clips_for_rank, rounds, run_sharded


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _clip_output(i):
    return torch.full((2, 3), float(i)) + torch.arange(6).reshape(2, 3) * 0.01


def _worker(rank, world, port, n_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ran = []

    def run_clip(i):
        ran.append(i)
        return _clip_output(i)

    res = run_sharded(n_clips, rank, world, dist, run_clip, lambda: torch.zeros(2, 3))
    ok = all(torch.equal(res[i], _clip_output(i)) for i in range(n_clips)) and len(res) == n_clips
    q.put((rank, ok, ran))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [4, 5, 1])
def test_sharded_run_world2(n_clips):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    [p.start() for p in ps]
    out = [q.get(timeout=120) for _ in range(world)]
    [p.join(60) for p in ps]
    assert all(ok for _, ok, _ in out)
    ran = sorted(i for _, _, r in out for i in r)
    assert ran == list(range(n_clips))                      # every clip computed exactly once
    for rank, _, r in out:
        assert r == clips_for_rank(n_clips, world, rank)    # clip i on rank i % world


def test_partition_helpers():
    assert clips_for_rank(10, 8, 1) == [1, 9] and clips_for_rank(3, 8, 5) == []
    assert rounds(17, 8) == 3 and rounds(8, 8) == 1
    got = run_sharded(3, 0, 1, None, lambda i: i * 10, lambda: -1)
    assert got == [0, 10, 20]
