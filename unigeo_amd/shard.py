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


def run_in_flight(n, nctx, run_clip, after_clip=None):
    """Several clips in flight on ONE GPU (round 5): clip ``i`` of this rank runs on context ``i % nctx``, every context on its own host thread
    (``run_clip(i, i % nctx)`` blocks until that clip is done - the engine call releases the GIL).  ``after_clip(i, i % nctx)`` is called on the CALLING thread,
    in clip order 0, 1, 2, ... - the place for the collective that reassembles a clip's output: every rank issues its collectives in the same order, from one
    thread - and context ``i % nctx`` does not start its next clip before ``after_clip(i, ...)`` has returned (the next run overwrites the context's outputs).
    Exceptions of a worker are re-raised here.  ``nctx == 1`` degenerates to the serial loop."""
    import threading
    nctx = max(1, min(nctx, n))
    if nctx == 1:
        for i in range(n):
            run_clip(i, 0)
            if after_clip:
                after_clip(i, 0)
        return
    done = [threading.Event() for _ in range(n)]
    released = [threading.Event() for _ in range(n)]
    errs = []

    def worker(j):
        try:
            for i in range(j, n, nctx):
                run_clip(i, j)
                done[i].set()
                if after_clip:
                    released[i].wait()
                if errs:
                    return
        except Exception as ex:       # surface on the calling thread
            errs.append(ex)
            for ev in done:
                ev.set()
    th = [threading.Thread(target=worker, args=(j,)) for j in range(nctx)]
    [t.start() for t in th]
    try:
        for i in range(n):
            done[i].wait()
            if errs:
                break
            if after_clip:
                after_clip(i, i % nctx)
                released[i].set()
    except Exception as ex:
        errs.append(ex)
    finally:
        for ev in released:
            ev.set()
        [t.join() for t in th]
    if errs:
        raise errs[0]


def _parse_cpulist(txt):
    out = []
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def host_topology(allowed=None, sysfs="/sys/devices/system"):
    """{numa node: [physical core = sorted list of its hyperthreads, ...]} restricted to the logical CPUs in ``allowed``.  Falls back to one node
    with every logical CPU as its own core where sysfs is not readable (containers without /sys)."""
    import glob
    import os
    allowed = sorted(allowed if allowed is not None else os.sched_getaffinity(0))
    aset = set(allowed)
    nodes = {}
    try:
        for nd in sorted(glob.glob(os.path.join(sysfs, "node", "node[0-9]*"))):
            cpus = [c for c in _parse_cpulist(open(os.path.join(nd, "cpulist")).read()) if c in aset]
            if cpus:
                nodes[int(os.path.basename(nd)[4:])] = cpus
    except OSError:
        nodes = {}
    if not nodes:
        nodes = {0: allowed}
    topo = {}
    for nd, cpus in nodes.items():
        seen, cores = set(), []
        for c in cpus:
            if c in seen:
                continue
            try:
                sib = [x for x in _parse_cpulist(open(os.path.join(sysfs, "cpu", f"cpu{c}", "topology", "thread_siblings_list")).read()) if x in aset]
            except OSError:
                sib = [c]
            sib = sorted(set(sib) | {c})
            seen.update(sib)
            cores.append(sib)
        topo[nd] = cores
    return topo


def gpu_numa_nodes(n):
    """NUMA node of each of the first ``n`` GPUs (HIP device order) from /sys/bus/pci/devices/<bus id>/numa_node, or None per device."""
    out = [None] * n
    try:
        import torch
        for i in range(min(n, torch.cuda.device_count())):
            pr = torch.cuda.get_device_properties(i)
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            try:
                v = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
                out[i] = v if v >= 0 else None
            except (OSError, ValueError):
                pass
    except Exception:
        pass
    return out


def plan_affinity(local_rank, local_world, topo, gpu_nodes=None):
    """Logical CPUs for ``local_rank`` of ``local_world`` ranks on this host: the rank is placed on its GPU's NUMA node (``gpu_nodes[local_rank]``; unknown ->
    the ranks are spread over the nodes in order), the ranks of one node split that node's PHYSICAL cores into contiguous slices, and a rank gets every
    hyperthread of its cores - two ranks never share a physical core unless there are more ranks than cores (slices of the sorted logical ids put
    rank r and rank r + 4 on SMT siblings of the same cores and ignored the GPU's socket)."""
    nodes = sorted(topo)
    def node_of(r):
        g = gpu_nodes[r] if gpu_nodes and r < len(gpu_nodes) else None
        return g if g in topo else nodes[(r * len(nodes)) // max(local_world, 1)]
    mine = node_of(local_rank)
    peers = [r for r in range(local_world) if node_of(r) == mine]
    cores = topo[mine]
    k, n = peers.index(local_rank), len(peers)
    if len(cores) >= n:
        lo, hi = (k * len(cores)) // n, ((k + 1) * len(cores)) // n
        sel = cores[lo:hi]
    else:
        sel = [cores[k % len(cores)]]
    return sorted(c for core in sel for c in core)


def pin_rank_to_cores(local_rank: int, world: int, local_world=None):
    """One process per GPU, and every process enqueues ~25 k kernel launches per clip from one host thread: with 8 ranks on one host the
    launch threads must not migrate onto each other's cores.  Gives the rank a slice of the PHYSICAL cores of its GPU's NUMA node (``plan_affinity``;
    the divisor is the number of ranks on THIS host - LOCAL_WORLD_SIZE - not the global world size); a single rank keeps what it has.  Returns the
    core list (for the bench line), or None where the platform has no affinity call.  eval.py of the reference is single-process
    (``/root/reference/eval.py:33-56``): there is nothing to mirror, this is launch-side hygiene of the sharded run."""
    import os
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    if world <= 1 or local_world <= 1 or not hasattr(os, "sched_getaffinity"):
        return None
    topo = host_topology(os.sched_getaffinity(0))
    mine = plan_affinity(local_rank % local_world, local_world, topo, gpu_numa_nodes(local_world))
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return mine
