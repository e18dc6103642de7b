"""Evaluation metrics and the CSV sink, restated from the reference:

* ``depth_evaluation``  <- ``/root/reference/metrics/eval_depth.py:6-246`` restricted to the call the harness makes
  (``eval.py:49``: ``align_with_lstsq=True`` + ``custom_mask``): mask 0 < gt < 80, least-squares scale/shift
  (``metrics/alignment.py:150-167``), then AbsRel / SqRel / RMSE / LogRMSE / delta thresholds on the custom mask.
* ``normal_evaluation`` <- ``/root/reference/metrics/eval_normal.py:4-72``: angular error in degrees, mean / median /
  rmse / % under 5, 7.5, 11.25, 22.5, 30 degrees.
* ``MetricsManager``    <- ``/root/reference/metrics/save_utils.py:5-90``: one row per sequence, NaN for missing
  metrics, trailing ``Average`` row (NaN-skipping mean), ``%.5f``.
Other alignment modes of the reference function are not used by the hot-path configs and raise here.
"""
import math
import os

import numpy as np


def _np(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


def depth_evaluation(predicted_depth_original, ground_truth_depth_original, max_depth=80, custom_mask=None,
                     align_with_lstsq=False, **unsupported):
    if not align_with_lstsq or any(v for v in unsupported.values()):
        raise NotImplementedError("only align_with_lstsq=True (the mode eval.py uses) is restated")
    pred = _np(predicted_depth_original).astype(np.float32).reshape(-1)
    gt = _np(ground_truth_depth_original).astype(np.float32).reshape(-1)
    cm = None if custom_mask is None else _np(custom_mask).astype(bool).reshape(-1)
    mask = (gt > 0) & (gt < max_depth) if max_depth is not None else gt > 0
    p, g = pred[mask], gt[mask]
    A = np.stack([p, np.ones_like(p)], 1)
    sol = np.linalg.lstsq(A, g[:, None], rcond=None)[0]
    s, t = np.float32(sol[0, 0]), np.float32(sol[1, 0])
    p = s * p + t
    if cm is not None:
        sel = cm[mask]
        p, g = p[sel], g[sel]
    n = int(p.size)
    if n == 0:
        vals = [0, 0, 0, 0, 0, 0, 0, 0]
    else:
        f32 = np.float32
        abs_rel = float(np.mean(np.abs(p - g) / g, dtype=f32))
        sq_rel = float(np.mean((p - g) ** 2 / g, dtype=f32))
        rmse = float(np.sqrt(np.mean((p - g) ** 2, dtype=f32)))
        pc = np.maximum(p, f32(1e-5))
        log_rmse = float(np.sqrt(np.mean((np.log(pc) - np.log(g)) ** 2, dtype=f32)))
        ratio = np.maximum(pc / g, g / pc)
        th = [float(np.mean((ratio < k).astype(f32))) for k in (1.0, 1.25, 1.25 ** 2, 1.25 ** 3)]
        vals = [abs_rel, sq_rel, rmse, log_rmse] + th
    keys = ["Abs Rel", "Sq Rel", "RMSE", "Log RMSE", "delta < 1.", "delta < 1.25", "delta < 1.25^2", "delta < 1.25^3"]
    res = dict(zip(keys, vals))
    res["valid_pixels"] = n
    return res, (float(s), float(t))


def normal_evaluation(predicted_normal_original, ground_truth_normal_original, custom_mask=None):
    pn = _np(predicted_normal_original).astype(np.float32)
    gn = _np(ground_truth_normal_original).astype(np.float32)
    dot = (pn * gn).sum(-1)
    cosang = dot / (np.linalg.norm(pn, axis=-1) * np.linalg.norm(gn, axis=-1) + np.float32(1e-6))
    err = np.degrees(np.arccos(np.clip(cosang, -1.0, 1.0))).astype(np.float32)
    e = err[_np(custom_mask).astype(bool)] if custom_mask is not None else err.reshape(-1)
    n = e.size
    if n == 0:   # empty mask: NaN metrics (the reference's mean over an empty selection); MetricsManager skips NaN rows
        return {k: float("nan") for k in ("normal mean", "normal median", "normal rmse", "angle < 5", "angle < 7.5",
                                          "angle < 11.25", "angle < 22.5", "angle < 30")}
    out = {"normal mean": float(e.mean(dtype=np.float32)), "normal median": float(np.sort(e)[(n - 1) // 2]),
           "normal rmse": float(np.sqrt((e * e).sum(dtype=np.float32) / n))}
    for k in (5, 7.5, 11.25, 22.5, 30):
        out[f"angle < {k:g}"] = float(100.0 * np.float32((e < k).sum()) / n)
    return out


class MetricsManager:
    def __init__(self, metric_names, sequence_names=None):
        self.metric_names = list(metric_names)
        self.sequence_names = [] if sequence_names is None else list(sequence_names)
        self.rows = {}

    def update_metrics(self, metrics_dict):
        seq = metrics_dict.get("seq_name")
        if seq is None:
            print("error: 'seq_name' missing from the metrics dict")
            return
        if seq not in self.rows:
            if seq not in self.sequence_names:
                self.sequence_names.append(seq)
            self.rows[seq] = {m: math.nan for m in self.metric_names}
        for m in self.metric_names:
            if m in metrics_dict:
                self.rows[seq][m] = float(metrics_dict[m])

    def calculate_averages(self):
        out = {}
        for m in self.metric_names:
            v = [r[m] for r in self.rows.values() if not math.isnan(r[m])]
            out[m] = sum(v) / len(v) if v else math.nan
        return out

    def export_to_csv(self, filepath):
        if not self.rows:
            print("warning: nothing to export")
            return
        d = os.path.dirname(filepath)
        if d:
            os.makedirs(d, exist_ok=True)
        fmt = lambda x: "" if math.isnan(x) else "%.5f" % x
        lines = ["," + ",".join(self.metric_names)]
        for seq in self.sequence_names:
            if seq in self.rows:
                lines.append(seq + "," + ",".join(fmt(self.rows[seq][m]) for m in self.metric_names))
        avg = self.calculate_averages()
        lines.append("Average," + ",".join(fmt(avg[m]) for m in self.metric_names))
        with open(filepath, "w") as f:
            f.write("\n".join(lines) + "\n")
        print(f"metrics export {filepath}")
