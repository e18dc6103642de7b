"""CPU: the harness loop (eval.py mirror) end to end with a stand-in model that answers from the GT."""
import numpy as np
import torch

from unigeo_amd.harness import SyntheticGeometryDataset, evaluate, parse_dataset_config, parse_metric_config, split_clips


class _GTModel:
    """Perfect predictor up to an affine depth ambiguity: exercises alignment + metrics."""

    def forward(self, data):
        d = np.stack([-np.asarray(c)[2] for c in data["cam_coord"]], 0)         # OpenGL z -> OpenCV depth
        n = np.stack([np.asarray(c).transpose(1, 2, 0) for c in data["cam_normal"]], 0)
        return {"pred_depths": torch.from_numpy(0.5 * d + 1.0).float(), "pred_normals": torch.from_numpy(n).float()}


CFG = {"dataset": "SyntheticGeometryDataset", "root": "unused", "h": 32, "w": 48, "clip_length": 5, "clip_overlap": 1,
       "model_name": "DepthCrafter", "model_params": {},
       "eval_depth": {"metric_names": ["Abs Rel", "delta < 1.25", "delta < 1.25^2", "delta < 1.25^3"], "depth_alignment": "lstsq"},
       "eval_normal": {"metric_names": ["normal mean", "normal median", "angle < 7.5", "angle < 11.25"]}}


def test_clip_split_rule():
    c = split_clips(12, 5, 1)
    assert list(c.keys()) == [0, 4, 8] and c[8] == [8, 9, 10, 11, 11] and c[0] == [0, 1, 2, 3, 4]
    assert all(len(v) == 25 for v in split_clips(45, 25, 5).values())


def test_config_parsing():
    assert parse_dataset_config(CFG) == {"root": "unused", "clip_length": 5, "clip_overlap": 1,
                                         "input_size": (32, 48), "target_size": (32, 48)}
    assert parse_metric_config(CFG)[:2] == ["Abs Rel", "delta < 1.25"] and len(parse_metric_config(CFG)) == 8


def test_evaluate_loop(tmp_path):
    ds = SyntheticGeometryDataset(**parse_dataset_config(CFG), num_frames=9)
    rows, mm = evaluate(CFG, dataset=ds, model=_GTModel(), save_dir=str(tmp_path), verbose=False)
    assert len(rows) == len(ds) == 3
    for r in rows:
        assert r["Abs Rel"] < 1e-5 and r["delta < 1.25"] == 1.0      # affine ambiguity removed by the lstsq alignment
        assert r["normal mean"] < 0.1
    txt = (tmp_path / "metrics.csv").read_text().strip().splitlines()
    assert txt[0].startswith(",Abs Rel,") and txt[-1].startswith("Average,") and len(txt) == 5
    # rank-sharded evaluation covers the same clips
    r0, _ = evaluate(CFG, dataset=ds, model=_GTModel(), save_dir=str(tmp_path / "r0"), rank=0, world=2, verbose=False)
    r1, _ = evaluate(CFG, dataset=ds, model=_GTModel(), save_dir=str(tmp_path / "r1"), rank=1, world=2, verbose=False)
    assert sorted(x["seq_name"] for x in r0 + r1) == sorted(x["seq_name"] for x in rows)
    # two plugin instances on one GPU, clips in flight on two host threads (round 5): same rows in the same order, same CSV
    ds5 = SyntheticGeometryDataset(**parse_dataset_config(CFG), num_frames=21)
    ra, _ = evaluate(CFG, dataset=ds5, model=_GTModel(), save_dir=str(tmp_path / "a"), verbose=False)
    rb, _ = evaluate(CFG, dataset=ds5, models=[_GTModel(), _GTModel()], save_dir=str(tmp_path / "b"), verbose=False)
    assert len(ra) == len(ds5) >= 5 and ra == rb
    assert (tmp_path / "a" / "metrics.csv").read_text() == (tmp_path / "b" / "metrics.csv").read_text()
