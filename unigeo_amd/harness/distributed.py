"""Rank-sharded evaluation (BASELINE config 3: full split, clips sharded over the 8 GPUs of a node).

Clips are independent samples in the reference (``/root/reference/eval.py:33-56``), so every rank runs the plain
per-clip loop on clips ``rank, rank+world, ...`` with its own model replica; the only exchange is the gather of the
per-clip metric rows (a few floats each) at the end, after which rank 0 writes the CSV in dataset order.  Works with
``nccl`` (= RCCL, one process per GPU) and with ``gloo`` (CPU tests).
"""
import os

from .eval import evaluate, parse_metric_config
from .metrics import MetricsManager


def evaluate_sharded(config, dataset, model, save_dir="./debug_output", dist=None, verbose=False):
    if dist is None or not dist.is_initialized():
        rank, world = 0, 1
    else:
        rank, world = dist.get_rank(), dist.get_world_size()
    rows, _ = evaluate(config, dataset=dataset, model=model, save_dir=os.path.join(save_dir, f"rank{rank}"),
                       rank=rank, world=world, verbose=verbose)
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, rows)
        rows = [r for part in gathered for r in part]
    rows.sort(key=lambda r: int(r["seq_name"].split("_", 1)[0]))   # "<data_idx>_<scene>": dataset order (numeric: >= 1000 clips)
    mm = MetricsManager(metric_names=parse_metric_config(config))
    for r in rows:
        mm.update_metrics(r)
    if rank == 0:
        mm.export_to_csv(os.path.join(save_dir, "metrics.csv"))
    return rows, mm
