"""CPU, world_size 2 (gloo): clip sharding + per-round all_gather reassembles outputs in dataset order,
including the ragged tail (n_clips % world != 0)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from unigeo_amd.shard import clips_for_rank, rounds, run_in_flight, run_sharded


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


def _worker_pinned(rank, world, port, n_clips, q):
    from unigeo_amd.shard import pin_rank_to_cores
    cores = pin_rank_to_cores(rank, world)
    _worker(rank, world, port, n_clips, q)
    q.put(("cores", rank, cores, sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None))


def test_sharded_run_world8_ragged_tail():
    """BASELINE configs[2] shape of the launch: 8 ranks, 17 clips (two full rounds + a tail round with ONE real clip and seven dummies), every
    rank pinned to its own slice of the host cores as bench.py / tools/eval_sharded.py do."""
    world, n_clips, port = 8, 17, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_pinned, args=(r, world, port, n_clips, q)) for r in range(world)]
    [p.start() for p in ps]
    got = [q.get(timeout=300) for _ in range(2 * world)]
    [p.join(60) for p in ps]
    out = [g for g in got if g[0] != "cores"]
    pins = {g[1]: (g[2], g[3]) for g in got if g[0] == "cores"}
    assert all(ok for _, ok, _ in out)
    assert sorted(i for _, _, r in out for i in r) == list(range(n_clips))
    for rank, _, r in out:
        assert r == clips_for_rank(n_clips, world, rank)
    assert sorted(len(r) for _, _, r in out) == [2] * 7 + [3]            # the tail round has one real clip
    for rank, (want, have) in pins.items():
        if want is not None:
            assert have == want and len(want) >= 1                         # the process really runs on the slice it was given
    slices = [tuple(w) for w, _ in pins.values() if w is not None]
    if len(slices) == world and len(set(c for s_ in slices for c in s_)) >= world:
        assert len(set(slices)) == world                                    # distinct slices when the host has >= world cores


def _inflight_worker(rank, world, port, n, nctx, q):
    """bench.py's per-rank protocol with several clips in flight: contexts = host threads with their own output buffer; a clip's output is all_gathered from the
    main thread, in clip order, before its context starts the next clip (which overwrites the buffer)."""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bufs = [torch.zeros(4) for _ in range(nctx)]          # one "depth buffer" per context, overwritten by every run
    rng = np.random.default_rng(rank)
    delay = rng.uniform(0.0, 0.02, n)                     # clips finish out of order across contexts, differently on every rank
    got, order = [], []

    def run_clip(i, j):
        time.sleep(float(delay[i]))
        bufs[j].fill_(1000.0 * rank + i)

    def after_clip(i, j):
        order.append(i)
        out = [torch.empty(4) for _ in range(world)]
        dist.all_gather(out, bufs[j])
        got.append([float(o[0]) for o in out])
    run_in_flight(n, nctx, run_clip, after_clip)
    q.put((rank, order, got))
    dist.destroy_process_group()


@pytest.mark.parametrize("n,nctx", [(7, 3), (4, 2), (3, 1), (2, 3)])
def test_clips_in_flight_keep_one_collective_order_world2(n, nctx):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_inflight_worker, args=(r, world, port, n, nctx, q)) for r in range(world)]
    [p.start() for p in ps]
    out = [q.get(timeout=120) for _ in range(world)]
    [p.join(60) for p in ps]
    for rank, order, got in out:
        assert order == list(range(n))                                             # collectives in clip order on every rank
        assert got == [[1000.0 * r + i for r in range(world)] for i in range(n)]   # clip i's gather saw clip i's output of every rank (no buffer overwritten early)


def test_in_flight_runner_propagates_worker_errors():
    def run_clip(i, j):
        if i == 2:
            raise ValueError("clip 2 failed")
    with pytest.raises(ValueError):
        run_in_flight(5, 2, run_clip, lambda i, j: None)


def test_affinity_plan_physical_cores_per_numa_node():
    """ranks must not land on SMT siblings of each other's cores, must sit on their GPU's socket, and the divisor is the number of
    ranks on THIS host.  Synthetic 2-socket host, 8 physical cores per socket, Linux SMT numbering (sibling of cpu c is c + 16)."""
    from unigeo_amd.shard import plan_affinity
    topo = {0: [[c, c + 16] for c in range(0, 8)], 1: [[c, c + 16] for c in range(8, 16)]}
    gpu_nodes = [0, 0, 1, 1, 1, 1, 0, 0]                                   # GPUs 2-5 hang off socket 1
    plans = [plan_affinity(r, 8, topo, gpu_nodes) for r in range(8)]
    assert all(len(p) == 4 for p in plans)                                  # 2 physical cores x 2 hyperthreads each
    assert len(set(c for p in plans for c in p)) == 32                      # disjoint, the whole host used
    for r, p in enumerate(plans):
        node_cpus = {c for core in topo[gpu_nodes[r]] for c in core}
        assert set(p) <= node_cpus                                          # on the GPU's own socket
        assert {c % 16 for c in p} == {c % 16 for c in p if c < 16}        # whole physical cores: both siblings or neither
    # unknown GPU placement: ranks spread over the sockets in order; more ranks than cores: still one core each
    assert plan_affinity(0, 2, topo) == sorted(c for core in topo[0] for c in core) and plan_affinity(1, 2, topo) == sorted(c for core in topo[1] for c in core)
    assert plan_affinity(5, 8, {0: [[0], [1]]}) == [1]


def test_partition_helpers():
    assert clips_for_rank(10, 8, 1) == [1, 9] and clips_for_rank(3, 8, 5) == []
    assert rounds(17, 8) == 3 and rounds(8, 8) == 1
    got = run_sharded(3, 0, 1, None, lambda i: i * 10, lambda: -1)
    assert got == [0, 10, 20]


def _eval_worker(rank, world, port, tmp, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_harness_cpu import CFG, _GTModel
    from unigeo_amd.harness import SyntheticGeometryDataset, parse_dataset_config
    from unigeo_amd.harness.distributed import evaluate_sharded
    ds = SyntheticGeometryDataset(**parse_dataset_config(CFG), num_frames=17)
    rows, mm = evaluate_sharded(CFG, ds, _GTModel(), save_dir=tmp, dist=dist)
    q.put((rank, [r["seq_name"] for r in rows], len(ds)))
    dist.destroy_process_group()


def test_sharded_evaluation_world2(tmp_path):
    """BASELINE config 3 plumbing: every rank evaluates its clips, rows are gathered, rank 0 writes one CSV."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_eval_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    [p.start() for p in ps]
    out = [q.get(timeout=180) for _ in range(world)]
    [p.join(60) for p in ps]
    n = out[0][2]
    for _, names, _ in out:
        assert names == sorted(names) and len(names) == n            # all clips, dataset order, on every rank
    lines = (tmp_path / "metrics.csv").read_text().strip().splitlines()
    assert len(lines) == n + 2 and lines[-1].startswith("Average,")
