#!/usr/bin/env python
"""python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/eval_sharded.py CONFIG.yaml
One process per GPU; clips sharded over ranks; rank 0 writes debug_output/metrics.csv.  UG_IN_FLIGHT=n (default 1): n plugin instances per GPU, this rank's
clips in flight on n host threads (round 5: +10 % aggregate frames/s at n = 2)."""
import os, sys
import yaml
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from unigeo_amd.harness import SyntheticGeometryDataset, parse_dataset_config, import_class_from_module
from unigeo_amd.harness.distributed import evaluate_sharded

cfg = yaml.safe_load(open(sys.argv[1]))
world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
multi = world > 1 or os.environ.get("UG_FORCE_DIST") == "1"      # UG_FORCE_DIST=1: take the RCCL path with a single rank (plumbing check)
from unigeo_amd.shard import pin_rank_to_cores
pin_rank_to_cores(local, world)                                  # each rank's launch thread keeps its own host cores
if multi:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29542")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(local)
    dist.init_process_group("nccl")
ds = import_class_from_module("unigeo_amd.harness", cfg["dataset"])(**parse_dataset_config(cfg))
model = import_class_from_module("unigeo_amd.model", cfg["model_name"])(device_id=local, **cfg["model_params"])
nfl = max(1, int(os.environ.get("UG_IN_FLIGHT", "1")))
models = [model] + [import_class_from_module("unigeo_amd.model", cfg["model_name"])(device_id=local, **cfg["model_params"]) for _ in range(nfl - 1)]
rows, mm = evaluate_sharded(cfg, ds, model, dist=dist if multi else None, verbose=(local == 0), models=models if nfl > 1 else None)
if multi:
    dist.destroy_process_group()
