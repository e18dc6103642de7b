"""Rank-sharded evaluation (BASELINE config 3: full split, clips sharded over the 8 GPUs of a node).

Clips are independent samples in the reference (``/root/reference/eval.py:33-56``), so every rank runs the plain
per-clip loop on clips ``rank, rank+world, ...`` with its own model replica; the only exchange is the gather of the
per-clip metric rows (a few floats each) at the end, after which rank 0 writes the CSV in dataset order.  Works with
``nccl`` (= RCCL, one process per GPU) and with ``gloo`` (CPU tests).
"""
import os
import time

from .eval import evaluate, parse_metric_config
from .metrics import MetricsManager


def evaluate_sharded(config, dataset, model, save_dir="./debug_output", dist=None, verbose=False, models=None):
    if dist is None or not dist.is_initialized():
        rank, world = 0, 1
    else:
        rank, world = dist.get_rank(), dist.get_world_size()
    t0 = time.perf_counter()
    rows, _ = evaluate(config, dataset=dataset, model=model, save_dir=os.path.join(save_dir, f"rank{rank}"),
                       rank=rank, world=world, verbose=verbose, models=models)     # models: several plugin instances on this rank's GPU, clips in flight (eval.py)
    t_eval = time.perf_counter() - t0
    timing = {"world": world, "per_rank": [{"rank": rank, "clips": len(rows), "eval_s": round(t_eval, 3)}], "gather_s": 0.0, "barrier_wait_s": 0.0}
    if world > 1 or (dist is not None and dist.is_initialized()):
        # efficiency breakdown of the sharded run: how long each rank computed, how long the fastest waited for the slowest, how long the
        # exchange itself took - so the first multi-GPU run yields the scaling analysis without a code change
        t1 = time.perf_counter()
        dist.barrier()
        t2 = time.perf_counter()
        gathered = [None] * world
        dist.all_gather_object(gathered, (rows, timing["per_rank"][0]))
        t3 = time.perf_counter()
        rows = [r for part, _ in gathered for r in part]
        timing = {"world": world, "per_rank": [info for _, info in gathered], "gather_s": round(t3 - t2, 4), "barrier_wait_s": round(t2 - t1, 4)}
    if rank == 0 and verbose:
        ev = [p["eval_s"] for p in timing["per_rank"]]
        print("[sharded] " + "  ".join(f"rank {p['rank']}: {p['clips']} clips {p['eval_s']:.2f} s" for p in timing["per_rank"]))
        print(f"[sharded] slowest rank {max(ev):.2f} s, mean {sum(ev) / len(ev):.2f} s (balance {sum(ev) / len(ev) / max(max(ev), 1e-9):.3f}); "
              f"rank 0 waited {timing['barrier_wait_s']:.3f} s at the barrier, metric-row gather {timing['gather_s']:.4f} s")
    evaluate_sharded.last_timing = timing
    rows.sort(key=lambda r: int(r["seq_name"].split("_", 1)[0]))   # "<data_idx>_<scene>": dataset order (numeric: >= 1000 clips)
    mm = MetricsManager(metric_names=parse_metric_config(config))
    for r in rows:
        mm.update_metrics(r)
    if rank == 0:
        mm.export_to_csv(os.path.join(save_dir, "metrics.csv"))
    return rows, mm
