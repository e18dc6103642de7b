"""The evaluation loop - mirrors ``/root/reference/eval.py:10-99`` and ``configs/config_utils.py:3-35``.

Differences kept deliberately small: the YAML path is an argument instead of being hard-coded (eval.py:11),
dataset / model classes can be passed as objects, visualisation and the point-cloud / camera-pose branches
(not produced by the DepthCrafter / StableNormal plugins) are not reproduced.
"""
import importlib
import os

from .io_utils import prepare_gt_label
from .metrics import MetricsManager, depth_evaluation, normal_evaluation


def import_class_from_module(module_name, class_name):
    return getattr(importlib.import_module(module_name), class_name)


def parse_dataset_config(config):
    out = {"root": config["root"], "clip_length": config.get("clip_length", 30),
           "clip_overlap": config.get("clip_overlap", 0), "input_size": (config["h"], config["w"]),
           "target_size": (config["h"], config["w"])}
    for k in ("split", "split_file", "scenes"):      # optional: which scene list the loader walks (default: the split file)
        if k in config:
            out[k] = config[k]
    return out


def parse_metric_config(config):
    names = []
    for key in ("eval_depth", "eval_pcd", "eval_camera", "eval_normal"):
        if key in config:
            names.extend(config[key]["metric_names"])
    return names


def evaluate(config, dataset=None, model=None, save_dir="./debug_output", rank=0, world=1, verbose=True,
             device_metrics=False, models=None):
    """Run the reference's per-clip loop.  With ``world > 1`` this rank only evaluates clips
    ``rank, rank+world, ...`` (clips are independent samples); rows are merged by the caller.
    ``device_metrics=True`` evaluates depth / normal metrics on the GPU against the outputs still resident in HBM
    (``ug_eval_depth`` / ``ug_eval_normal``) instead of on the host copies.
    ``models`` = several instances of the plugin on ONE GPU (round 5): this rank's k-th clip runs on ``models[k % len(models)]``, each instance on its own host
    thread - independent clips in flight on one GPU, the sharding over GPUs one level down (+10 % aggregate frames/s with two DepthCrafter contexts: a second
    clip fills the CUs that one clip's tile tails and under-filled launches leave idle).  Rows / CSV come out in dataset order, identical to the serial loop."""
    if dataset is None:
        dataset = import_class_from_module("unigeo_amd.harness", config["dataset"])(**parse_dataset_config(config))
    if model is None and not models:
        model = import_class_from_module("unigeo_amd.model", config["model_name"])(**config["model_params"])
    mm = MetricsManager(metric_names=parse_metric_config(config))
    os.makedirs(save_dir, exist_ok=True)
    save_path = os.path.join(save_dir, "metrics.csv")
    def one_clip(data_idx, mdl):
        data = dataset[data_idx]
        seq = f"{data_idx:03d}_{data['scene_name']}"
        if verbose:
            print("processing seq:", seq)
        output = mdl.forward(data)
        gt = prepare_gt_label(data)
        metric = {"seq_name": seq}
        eng = getattr(getattr(mdl, "pipeline", None), "engine", None) if device_metrics else None
        if "eval_depth" in config:
            if eng is not None:
                res = eng.eval_depth(gt["gt_depths"].numpy(), gt["gt_masks"].numpy())
            else:
                res = depth_evaluation(output["pred_depths"], gt["gt_depths"], custom_mask=gt["gt_masks"], align_with_lstsq=True)
            metric.update(res[0])
        if "eval_normal" in config:
            if eng is not None:
                metric.update(eng.eval_normal(gt["gt_normals"].numpy(), gt["gt_masks"].numpy()))
            else:
                metric.update(normal_evaluation(output["pred_normals"], gt["gt_normals"], custom_mask=gt["gt_masks"]))
        return metric

    mine = list(range(rank, len(dataset), world))
    rows = []
    if models and len(models) > 1:
        import threading
        out, errs = [None] * len(mine), []

        def worker(j):
            try:
                for k in range(j, len(mine), len(models)):
                    out[k] = one_clip(mine[k], models[j])
            except Exception as ex:
                errs.append(ex)
        th = [threading.Thread(target=worker, args=(j,)) for j in range(len(models))]
        [t.start() for t in th]; [t.join() for t in th]
        if errs:
            raise errs[0]
        for metric in out:                     # dataset order, as the serial loop writes them
            rows.append(metric)
            mm.update_metrics(metric)
        mm.export_to_csv(save_path)
        return rows, mm
    for data_idx in mine:
        metric = one_clip(data_idx, models[0] if models else model)
        rows.append(metric)
        mm.update_metrics(metric)
        mm.export_to_csv(save_path)
    return rows, mm
