"""Clip sharding over the GPUs of one node.

Each dataset sample is one independent clip (the reference evaluates clips one by one in
``/root/reference/eval.py:33-56`` and never stitches overlaps), so the path shards with NO data-path
collective: clip ``i`` runs on rank ``i % world``, every rank holds a full weight replica.  The only
exchange is the reassembly of per-clip outputs in dataset order: one all_gather (RCCL over xGMI when the
tensors are HIP memory, gloo on CPU in the tests) per round of ``world`` clips.
"""
import numpy as np


def clips_for_rank(n_clips: int, world: int, rank: int):
    return list(range(rank, n_clips, world))


def rounds(n_clips: int, world: int):
    return (n_clips + world - 1) // world


class DeviceArray:
    """Minimal ``__cuda_array_interface__`` carrier so torch can view engine-owned HIP memory zero-copy."""

    def __init__(self, ptr, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def gather_round(local, dist, world):
    """all_gather one per-rank tensor (same shape on every rank) -> list ordered by rank."""
    import torch
    out = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(out, local.contiguous())
    return out


def run_sharded(n_clips, rank, world, dist, run_clip, make_dummy):
    """Run ``run_clip(i) -> tensor`` for this rank's clips; after every round all ranks exchange their
    result; returns the list of all clip outputs in dataset order (on every rank).  Ranks without a clip
    in the tail round contribute ``make_dummy()`` (dropped on reassembly)."""
    mine = clips_for_rank(n_clips, world, rank)
    results = [None] * n_clips
    for r in range(rounds(n_clips, world)):
        local = run_clip(mine[r]) if r < len(mine) else make_dummy()
        if world == 1:
            got = [local]
        else:
            got = gather_round(local, dist, world)
        for src in range(world):
            i = r * world + src
            if i < n_clips:
                results[i] = got[src]
    return results


def pin_rank_to_cores(local_rank: int, world: int):
    """One process per GPU, and every process enqueues ~25 k kernel launches per clip from one host thread: with 8 ranks on one host the
    launch threads must not migrate onto each other's cores.  Gives rank ``local_rank`` the ``local_rank``-th contiguous slice of the cores
    this process is allowed to use (``os.sched_getaffinity``), at least one core; a single rank keeps what it has.  Returns the core list
    (for the bench line), or None where the platform has no affinity call.  eval.py of the reference is single-process
    (``/root/reference/eval.py:33-56``): there is nothing to mirror, this is launch-side hygiene of the sharded run."""
    import os
    if world <= 1 or not hasattr(os, "sched_getaffinity"):
        return None
    cores = sorted(os.sched_getaffinity(0))
    per = max(1, len(cores) // world)
    lo = (local_rank * per) % len(cores)
    mine = cores[lo:lo + per] or cores[:1]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return mine
