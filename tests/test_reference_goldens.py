"""CPU: the oracle and the host-side mirrors against golden vectors produced by the reference's OWN Python
(tests/golden/make_goldens.py ran /root/reference code in the build container; only the .npz/.csv travel)."""
import io
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_goldens.npz"), allow_pickle=False)


def test_g1_prepare_input_uint8_truncation():
    from oracle.pipeline import prepare_input
    from unigeo_amd.model.depthcrafter import DepthCrafter
    imgs = list(G["g1_images"])
    np.testing.assert_array_equal(prepare_input(imgs), G["g1_frames"])
    np.testing.assert_array_equal(DepthCrafter.prepare_input(None, {"images": imgs}), G["g1_frames"])


def test_g2_depth_postprocessing():
    from oracle.pipeline import depth_from_frames
    got = np.stack(depth_from_frames(G["g2_res"].copy()))
    np.testing.assert_array_equal(got, G["g2_depths"])
    assert got.min() >= 1 / 1.1 - 1e-6 and got.max() <= 10.0 + 1e-6


def test_g3_surface_normals_oracle_matches_reference():
    from oracle.geometry import prepare_output
    d, n = prepare_output(list(G["g3_depths"]), list(G["g3_K"]))
    np.testing.assert_array_equal(d.numpy(), G["g3_pred_depths"])
    ref = G["g3_pred_normals"]
    cos = np.clip((n.numpy() * ref).sum(-1), -1, 1)
    ang = np.degrees(np.arccos(cos))
    # same algorithm, same fp32 lstsq: identical up to LAPACK batching order
    assert ang.max() < 0.05, ang.max()


def test_g3_surface_normals_oracle_matches_reference_full_frame():
    """One 384x512 frame: the oracle against the reference's own prepare_output output (reference_g3_fullframe.npz, made by make_goldens.py)."""
    from oracle.geometry import prepare_output
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_g3_fullframe.npz"))
    d, n = prepare_output([g["g3f_depth"]], [g["g3f_K"]])
    np.testing.assert_array_equal(d.numpy(), g["g3f_pred_depths"])
    ang = np.degrees(np.arccos(np.clip((n.numpy() * g["g3f_pred_normals"]).sum(-1), -1, 1)))
    assert ang.mean() < 0.02 and ang.max() < 0.2, (ang.mean(), ang.max())      # measured 0.007 / 0.044 degrees: the fp32 lstsq's batching order


def test_g4_prepare_gt_label():
    from unigeo_amd.harness import prepare_gt_label
    data = {k[len("g4_in_"):]: list(G[k]) for k in G.files if k.startswith("g4_in_")}
    out = prepare_gt_label(data)
    for k in ("gt_world_pts", "gt_masks", "gt_poses", "gt_depths", "gt_rgbs", "gt_normals"):
        ref = G["g4_out_" + k]
        got = out[k].numpy()
        assert got.shape == ref.shape and got.dtype == ref.dtype, k
        np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-6, err_msg=k)


def test_g5_depth_and_normal_metrics():
    from unigeo_amd.harness import depth_evaluation, normal_evaluation
    res = depth_evaluation(G["g5_pred_d"], G["g5_gt_d"], custom_mask=G["g5_mask"], align_with_lstsq=True)[0]
    for k, v in zip(G["g5_depth_keys"], G["g5_depth_vals"]):
        assert res[str(k)] == pytest.approx(float(v), rel=2e-5, abs=1e-6), k
    nres = normal_evaluation(G["g5_pred_n"], G["g5_gt_n"], custom_mask=G["g5_mask"])
    for k, v in zip(G["g5_normal_keys"], G["g5_normal_vals"]):
        assert nres[str(k)] == pytest.approx(float(v), rel=2e-5, abs=1e-4), k


def test_g6_metrics_manager_csv(tmp_path):
    from unigeo_amd.harness import MetricsManager
    mm = MetricsManager(metric_names=["Abs Rel", "delta < 1.25", "normal mean", "angle < 11.25"])
    rows = [{"seq_name": "000_sceneA", "Abs Rel": 0.123456789, "delta < 1.25": 0.87654321, "normal mean": 21.5, "angle < 11.25": 33.333333},
            {"seq_name": "001_sceneB", "Abs Rel": 0.2, "delta < 1.25": 0.75, "normal mean": 19.25}]
    p = tmp_path / "out" / "metrics.csv"
    for r in rows:
        mm.update_metrics(r)
        mm.export_to_csv(str(p))
    ref = open(os.path.join(os.path.dirname(__file__), "golden", "metrics_manager.csv")).read()
    assert p.read_text() == ref


def test_g7_stablenormal_uint8_wrap():
    from oracle.pipeline import stablenormal_post
    from unigeo_amd.model.stablenormal import StableNormal
    imgs = list(G["g7_in"])
    n, d = stablenormal_post([i.copy() for i in imgs])
    np.testing.assert_array_equal(n.numpy(), G["g7_normals"])
    out = StableNormal.postprocess([i.copy() for i in imgs])
    np.testing.assert_array_equal(out["pred_normals"].numpy(), G["g7_normals"])
    assert out["pred_depths"].shape == (2, 6, 8) and float(out["pred_depths"].abs().max()) == 0.0
    plug = StableNormal(predictor=lambda im: np.array(im))
    res = plug.forward({"images": [np.transpose(i, (2, 0, 1)).astype(np.float32) for i in imgs]})
    np.testing.assert_array_equal(res["pred_normals"].numpy(), G["g7_normals"])
    with pytest.raises(FileNotFoundError):       # no checkpoints, no synthetic_weights=True, no predictor: refused, never faked
        StableNormal()
