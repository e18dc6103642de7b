"""GPU: plugin surface, on-device normals, multi-GPU plumbing and full-size invariants, through the C ABI."""
import os

import numpy as np
import pytest
import torch

from util import assert_close

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_goldens.npz"), allow_pickle=False)


def _angle(a, b):
    return np.degrees(np.arccos(np.clip((a * b).sum(-1), -1, 1)))


def test_normals_kernel_vs_reference_golden(engine):
    """G3: the reference's own prepare_output (lstsq in fp32).  The kernel solves the same 3x3 systems in fp64;
    the reference's fp32 solve is itself noisy at cond(A^T A) ~ 1e5, so parity is on the angle: mean < 0.05 deg,
    99.9th percentile < 1 deg, every normal unit length and facing the camera."""
    got = engine.normals_from_depth(G["g3_depths"], G["g3_K"])
    ref = G["g3_pred_normals"]
    ang = _angle(got, ref)
    assert np.isfinite(got).all()
    assert ang.mean() < 0.05 and np.percentile(ang, 99.9) < 1.0, (ang.mean(), ang.max())
    np.testing.assert_allclose(np.linalg.norm(got, axis=-1), 1.0, atol=1e-5)


def _exact_normals_fp64(d, K):
    """The reference's normal equations ((A^T A + 1e-6 I) n = A^T 1 from 5x5 zero-padded box moments of the fp32 camera points, utils/geometry_utils.py:9-70)
    evaluated in float64 with a direct solve: the exact solution of the system the reference solves in fp32 (test infrastructure, CPU)."""
    import torch.nn.functional as F
    from oracle.geometry import backproject
    xyz = torch.from_numpy(backproject(d, K)).float().double()
    p = xyz.permute(2, 0, 1)[None]
    x, y, z = p[:, 0:1], p[:, 1:2], p[:, 2:3]
    k = torch.ones(1, 1, 5, 5, dtype=torch.float64)
    box = lambda t: F.conv2d(t, k, padding=2)[0, 0]
    ata = torch.stack([box(x * x), box(x * y), box(x * z), box(x * y), box(y * y), box(y * z), box(x * z), box(y * z), box(z * z)], -1).reshape(*xyz.shape[:2], 3, 3)
    n = torch.linalg.solve(ata + 1e-6 * torch.eye(3, dtype=torch.float64), torch.stack([box(x), box(y), box(z)], -1)[..., None])[..., 0]
    n = n / n.norm(dim=-1, keepdim=True)
    n[(n * xyz).sum(-1) > 0] *= -1
    n[:, :, 1:] = -n[:, :, 1:]
    return n.float().numpy()


def test_normals_kernel_vs_reference_golden_full_frame(engine):
    """G3 at full frame size (VERDICT r5 missing 4): one 384x512 frame through the REFERENCE's own prepare_output (back-projection + get_surface_normal
    + y/z flip, /root/reference/utils/geometry_utils.py:9-70,246-253, model/depthcrafter.py:48-59; fixture tests/golden/reference_g3_fullframe.npz made by
    make_goldens.py) against k_normals.
    What the fixture shows (round 6, measured): the reference forms the UNCENTRED 5x5 moments and solves in fp32 - at depth ~3 and a patch 0.04 wide the moments
    cancel to ~1e-4 of their size, so its own rounding moves its normals by 0.19 degrees on average (3.7 max) away from the EXACT solution of its own system
    (float64, below).  k_normals solves in float64: it sits on the exact solution (< 0.01 degrees) and therefore exactly that far from the fixture; the oracle, which
    calls the same fp32 torch routines as the reference, reproduces the fixture's noise to 0.007 / 0.044 degrees (tests/test_reference_goldens.py).  No independent
    implementation can be closer to the fixture than the fixture is to its exact solution; both distances are asserted."""
    from util import report
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_g3_fullframe.npz"))
    got = engine.normals_from_depth(g["g3f_depth"][None], g["g3f_K"][None])
    ref = g["g3f_pred_normals"][0]
    exact = _exact_normals_fp64(g["g3f_depth"], g["g3f_K"])
    a_ref, a_exact, a_noise = _angle(got[0], ref), _angle(got[0], exact), _angle(exact, ref)
    for nm, a in (("kernel_vs_reference", a_ref), ("kernel_vs_exact_fp64", a_exact), ("reference_vs_its_exact_solution", a_noise)):
        report(f"normals.full_frame.{nm}.mean_deg", float(a.mean())); report(f"normals.full_frame.{nm}.p999_deg", float(np.percentile(a, 99.9)))
        report(f"normals.full_frame.{nm}.max_deg", float(a.max()))
    assert np.isfinite(got).all()
    assert a_exact.mean() < 0.01 and a_exact.max() < 0.1, (a_exact.mean(), a_exact.max())               # the kernel IS the exact solution of the reference's system (measured 0.0035 / 0.028: one fp32 ulp of the cosine is 0.028 degrees)
    assert a_ref.mean() < 1.15 * a_noise.mean() + 0.01 and np.percentile(a_ref, 99.9) < 1.15 * np.percentile(a_noise, 99.9) + 0.05, (a_ref.mean(), a_noise.mean())
    assert a_ref.mean() < 0.3 and np.percentile(a_ref, 99.9) < 2.0 and a_ref.max() < 6.0, (a_ref.mean(), np.percentile(a_ref, 99.9), a_ref.max())   # measured 0.188 / 1.27 / 3.74
    np.testing.assert_allclose(np.linalg.norm(got, axis=-1), 1.0, atol=1e-5)


def test_normals_kernel_vs_oracle_full_frame(engine):
    from oracle.geometry import prepare_output
    from unigeo_amd.synthetic import synthetic_clip
    H, W = 384, 512
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    d = (3.0 + np.sin(xx / 50.0) * np.cos(yy / 37.0) + 0.003 * yy).astype(np.float32)
    K = synthetic_clip(1, H, W)["intrinsics"][0]
    got = engine.normals_from_depth(d[None], K[None])
    _, ref = prepare_output([d], [K])
    ang = _angle(got[0], ref[0].numpy())
    assert ang.mean() < 0.2 and np.percentile(ang, 99) < 2.0, (ang.mean(), np.percentile(ang, 99))


@pytest.fixture(scope="module")
def tiny_plugin():
    from unigeo_amd import weights as W
    from unigeo_amd.model import DepthCrafter
    m = DepthCrafter(synthetic_weights=True, cfgs=W.tiny_cfgs(), num_inference_steps=2, workspace_bytes=3 << 30)
    yield m
    m.pipeline.engine.close()


def test_plugin_contract(tiny_plugin):
    """forward(data) -> {'pred_depths' [Nf,H,W], 'pred_normals' [Nf,H,W,3]} CPU float32 tensors
    (reference model/depthcrafter.py:62-68), depth range of 1/(x+0.1) with x in [0,1]."""
    from unigeo_amd.synthetic import synthetic_clip
    data = synthetic_clip(3, 64, 128, seed=3)
    data["_index"] = 7                                    # per-clip noise seed = plugin seed + dataset index
    out = tiny_plugin.forward(data)
    d, n = out["pred_depths"], out["pred_normals"]
    assert d.shape == (3, 64, 128) and n.shape == (3, 64, 128, 3)
    assert d.dtype == torch.float32 and n.dtype == torch.float32 and d.device.type == "cpu"
    assert torch.isfinite(d).all() and torch.isfinite(n).all()
    assert float(d.min()) >= 1 / 1.1 - 1e-5 and float(d.max()) <= 10.0 + 1e-4
    assert float(d.max()) == pytest.approx(10.0, rel=1e-5) and float(d.min()) == pytest.approx(1 / 1.1, rel=1e-5)
    np.testing.assert_allclose(n.norm(dim=-1).numpy(), 1.0, atol=1e-4)
    out2 = tiny_plugin.forward(data)                      # same clip index -> same noise -> bit-identical (deterministic kernels)
    assert torch.equal(out2["pred_depths"], d) and torch.equal(out2["pred_normals"], n)
    # prepare_output keeps the reference's (depths, data) signature (model/depthcrafter.py:48) and reproduces forward's normals
    po = tiny_plugin.prepare_output(list(d.numpy()), data)
    assert torch.equal(po["pred_depths"], d) and torch.allclose(po["pred_normals"], n, atol=1e-6)
    data["_index"] = 8                                    # another clip draws independent noise (reference: fresh RNG draws)
    assert not torch.equal(tiny_plugin.forward(data)["pred_depths"], d)
    with pytest.raises(ValueError):
        tiny_plugin.forward(synthetic_clip(2, 60, 64))    # not a multiple of 64: rejected, not padded


def test_harness_end_to_end_with_plugin(tiny_plugin, tmp_path):
    from unigeo_amd.harness import SyntheticGeometryDataset, evaluate
    cfg = {"root": "x", "h": 64, "w": 64, "clip_length": 3, "clip_overlap": 1,
           "eval_depth": {"metric_names": ["Abs Rel", "delta < 1.25"]}, "eval_normal": {"metric_names": ["normal mean"]}}
    ds = SyntheticGeometryDataset(clip_length=3, clip_overlap=1, input_size=(64, 64), num_frames=5)
    rows, _ = evaluate(cfg, dataset=ds, model=tiny_plugin, save_dir=str(tmp_path), verbose=False)
    assert len(rows) == 3 and all(np.isfinite(r["Abs Rel"]) and np.isfinite(r["normal mean"]) for r in rows)
    assert (tmp_path / "metrics.csv").exists()


def test_harness_clips_in_flight_equal_the_serial_loop_on_the_gpu(tiny_plugin, tmp_path):
    """VERDICT r5 missing 2: harness.evaluate(models=[a, b, c]) - three plugin instances on ONE GPU, clips in flight on three host threads - against the serial
    loop of the reference harness (eval.py:33-39) on the GPU: the same rows (every metric equal bit for bit - the clip's noise seed comes from the dataset index,
    the kernels are deterministic and contexts share nothing but the chip), the same CSV."""
    from unigeo_amd import weights as W
    from unigeo_amd.harness import SyntheticGeometryDataset, evaluate
    from unigeo_amd.model import DepthCrafter
    cfg = {"root": "x", "h": 64, "w": 64, "clip_length": 3, "clip_overlap": 1,
           "eval_depth": {"metric_names": ["Abs Rel", "delta < 1.25"]}, "eval_normal": {"metric_names": ["normal mean", "normal median"]}}
    ds = SyntheticGeometryDataset(clip_length=3, clip_overlap=1, input_size=(64, 64), num_frames=15)
    extra = [DepthCrafter(synthetic_weights=True, cfgs=W.tiny_cfgs(), num_inference_steps=2, workspace_bytes=3 << 30) for _ in range(2)]
    try:
        for m in [tiny_plugin] + extra:
            m.pipeline.engine.set_coscheduled(True)                      # the same heuristics in both loops: the comparison is about sharing the GPU, nothing else
        serial, _ = evaluate(cfg, dataset=ds, model=tiny_plugin, save_dir=str(tmp_path / "s"), verbose=False)
        flight, _ = evaluate(cfg, dataset=ds, models=[tiny_plugin] + extra, save_dir=str(tmp_path / "f"), verbose=False)
        flight_dev, _ = evaluate(cfg, dataset=ds, models=[tiny_plugin] + extra, save_dir=str(tmp_path / "fd"), verbose=False, device_metrics=True)
    finally:
        tiny_plugin.pipeline.engine.set_coscheduled(False)
        for m in extra:
            m.pipeline.engine.close()
    assert len(serial) == len(ds) >= 7 and [r["seq_name"] for r in flight] == [r["seq_name"] for r in serial]
    assert flight == serial, "clips in flight changed a metric"
    assert (tmp_path / "s" / "metrics.csv").read_text() == (tmp_path / "f" / "metrics.csv").read_text()
    for a_, b_ in zip(serial, flight_dev):
        for k in ("Abs Rel", "delta < 1.25", "normal mean", "normal median"):
            assert b_[k] == pytest.approx(a_[k], rel=2e-4, abs=1e-3), k


def test_scannetpp_layout_to_metrics_with_plugin(tiny_plugin, tmp_path):
    """The whole chain of configs/depthcrafter_scannetpp.yaml on files in the processed ScanNet++ layout: loader (resized to a
    legal network size) -> plugin on the GPU -> GT preparation -> lstsq-aligned depth / normal metrics -> CSV."""
    import os
    from unigeo_amd.harness import evaluate
    root = os.path.join(os.path.dirname(__file__), "golden", "scannetpp_scene")
    cfg = {"dataset": "ScannetPPDataset", "root": root, "h": 64, "w": 64, "clip_length": 3, "clip_overlap": 1, "split": "test", "scenes": "all",
           "eval_depth": {"metric_names": ["Abs Rel", "delta < 1.25"], "depth_alignment": "lstsq"},
           "eval_normal": {"metric_names": ["normal mean", "angle < 11.25"]}}
    rows, _ = evaluate(cfg, model=tiny_plugin, save_dir=str(tmp_path), verbose=False)
    assert [r["seq_name"] for r in rows] == ["000_sceneA", "001_sceneA"]
    assert all(np.isfinite(r["Abs Rel"]) and 0 <= r["delta < 1.25"] <= 1 and np.isfinite(r["normal mean"]) for r in rows)
    rows_dev, _ = evaluate(cfg, model=tiny_plugin, save_dir=str(tmp_path / "dev"), verbose=False, device_metrics=True)
    for a_, b_ in zip(rows, rows_dev):                          # on-device metrics agree with the host mirror
        assert a_["Abs Rel"] == pytest.approx(b_["Abs Rel"], rel=2e-3, abs=1e-4)
        assert a_["normal mean"] == pytest.approx(b_["normal mean"], rel=2e-3, abs=1e-3)


def test_rccl_gather_of_engine_memory_zero_copy(tiny_plugin):
    """The N>1 path hands engine-owned HIP memory to torch.distributed (RCCL).  With one GPU this checks the
    zero-copy view and a world-size-1 nccl all_gather; the sharding logic itself is covered on CPU/gloo."""
    import torch.distributed as dist
    from unigeo_amd.shard import DeviceArray
    from unigeo_amd.synthetic import synthetic_clip
    tiny_plugin.forward(synthetic_clip(3, 64, 64, seed=1))
    eng = tiny_plugin.pipeline.engine
    ptr, shape = eng.device_ptrs()["depth"]
    view = torch.as_tensor(DeviceArray(ptr, shape), device="cuda:0")
    assert view.data_ptr() == ptr and tuple(view.shape) == shape
    _, host, _ = eng.get_outputs(frames=False, depth=True)
    assert np.array_equal(view.cpu().numpy(), host)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        out = [torch.empty_like(view)]
        dist.all_gather(out, view)
        torch.cuda.synchronize()
        assert torch.equal(out[0], view)
    finally:
        dist.destroy_process_group()


def test_full_size_invariants():
    """BASELINE configs[1] geometry (25 x 384 x 512) with 1 Euler step: size-independent properties -
    finite, in range, bit-reproducible run to run, and VAE-encode chunking changes the result by fp16 ulps only."""
    from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
    from unigeo_amd.synthetic import synthetic_clip
    from unigeo_amd.model.depthcrafter import DepthCrafter
    pipe = DepthCrafterPipelineHIP.from_random(seed=42, workspace_bytes=40 << 30)
    try:
        T, H, W = 25, 384, 512
        clip = synthetic_clip(T, H, W)
        frames = DepthCrafter.prepare_input(None, clip)
        nl, na = make_noise(T, H, W, 0)
        r1 = pipe(frames, num_inference_steps=1, window_size=T, noise_latents=nl, noise_aug=na)
        f1, d1 = r1.frames[0].copy(), r1.depth.copy()
        assert f1.shape == (T, H, W, 3) and np.isfinite(f1).all() and f1.min() >= 0 and f1.max() <= 1
        assert d1.min() >= 1 / 1.1 - 1e-5 and d1.max() <= 10 + 1e-4
        r2 = pipe(frames, num_inference_steps=1, window_size=T, noise_latents=nl, noise_aug=na)
        assert np.array_equal(r2.frames[0], f1) and np.array_equal(r2.depth, d1)
        # the VAE encode / decode chunks and the CLIP tower on 2 / 3 concurrent HIP streams (ug_set_concurrency): same kernels, same launch
        # parameters, disjoint outputs - not a bit may change
        for lanes in (2, 3):
            pipe.engine.set_concurrency(lanes)
            r3 = pipe(frames, num_inference_steps=1, window_size=T, noise_latents=nl, noise_aug=na)
            assert np.array_equal(r3.frames[0], f1) and np.array_equal(r3.depth, d1), f"{lanes} lanes changed the result"
        pipe.engine.set_concurrency(1)
        eng = pipe.engine
        v = (frames[:9] * 2 - 1).astype(np.float32)
        a = eng.vae_encode(v)
        b = np.concatenate([eng.vae_encode(v[:4]), eng.vae_encode(v[4:])], 0)
        # frames are independent in the encoder; chunking only changes reduction orders (GroupNorm partial sums,
        # split-K), i.e. a few fp16 ulps - not bit-exact, exactly like cuDNN algorithm changes in the reference
        assert_close(a, b, 5e-3, "VAE encode chunk invariance")
    finally:
        pipe.engine.close()


def test_full_size_unet_fused_feed_forward_paths_agree():
    """The fused GEGLU feed-forward kernel, its in-kernel LayerNorm and the whole-rounds row split only engage from M = 32768 token rows
    (level 0 of the 25 x 384 x 512 clip), beyond what the CPU oracle can reach.  The two-GEMM + LayerNorm-launch path they replace is the one
    every oracle-parity test (tiny and full-architecture configs) exercises, so at full size the UNet must give the same answer either way:
    * fused kernel (+ row split) vs two launches: BIT-IDENTICAL - same MFMA chain order, same fp16 roundings of the intermediate;
    * with the LayerNorm inside the kernel the fp32 row sums are taken in another order, which flips the fp16 rounding of ~0.15 % of the
      normalised values (1.7 % of rows, <= 2.7e-4 at the op level); through the whole UNet that grows to 2.1e-3 of max|v| (measured) -
      the same size as any other reordering here (VAE-encode chunking: bound 5e-3).  Bound 4e-3."""
    from unigeo_amd.pipeline import DepthCrafterPipelineHIP
    pipe = DepthCrafterPipelineHIP.from_random(seed=42, workspace_bytes=40 << 30)
    try:
        eng = pipe.engine
        T, h, w = 25, 48, 64
        rng = np.random.default_rng(5)
        x = rng.standard_normal((T, eng.unet_cfg.in_channels, h, w)).astype(np.float32)
        emb = rng.standard_normal((T, eng.unet_cfg.cross_attention_dim)).astype(np.float32)
        outs = {}
        for name, (on, pre) in {"two launches": (False, False), "fused": (True, False), "fused + in-kernel LayerNorm": (True, True)}.items():
            eng.set_ff_fused(on, prenorm=pre)
            outs[name] = eng.unet_forward(x, 1.2, emb)
        eng.set_ff_fused(True, prenorm=True)
        ref = outs["two launches"]
        assert np.isfinite(ref).all()
        assert np.array_equal(outs["fused"], ref), "fused feed-forward (whole rounds fused, thin tail through two GEMMs) is not bit-identical to two launches"
        # the level-0 3x3 convolutions run as two row-range launches (whole rounds on 256x160 tiles + the rest; launch_gemm): same MFMA chain
        # per output element, so switching the split off (knob 1024) must not change a bit - residuals, time-embedding bias and all
        try:
            eng.set_ff_fused(False, prenorm=False)
            eng.tune_force(-100 - 1024, 0)
            nosplit = eng.unet_forward(x, 1.2, emb)
        finally:
            eng.tune_force(-100, 0)
            eng.set_ff_fused(True, prenorm=True)
        assert np.array_equal(nosplit, ref), "row-split 3x3 convolutions are not bit-identical to single launches"
        assert_close(outs["fused + in-kernel LayerNorm"], ref, 4e-3, "full-size UNet forward, in-kernel LayerNorm vs LayerNorm launches")
    finally:
        pipe.engine.close()


def test_larger_clip_geometry():
    """BASELINE configs[4] geometry in fp16 (50 frames at 576 x 768, 1 Euler step): T = 50 temporal attention / pooled GroupNorm,
    S = 6912 spatial attention, > 2^31-element-free 32-bit buffer offsets, 12 GiB of activations - finite, in range, reproducible."""
    from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
    from unigeo_amd.synthetic import synthetic_clip
    from unigeo_amd.model.depthcrafter import DepthCrafter
    pipe = DepthCrafterPipelineHIP.from_random(seed=42, workspace_bytes=64 << 30)
    try:
        T, H, W = 50, 576, 768
        clip = synthetic_clip(T, H, W)
        frames = DepthCrafter.prepare_input(None, clip)
        nl, na = make_noise(T, H, W, 0)
        r1 = pipe(frames, num_inference_steps=1, window_size=T, noise_latents=nl, noise_aug=na)
        d1 = r1.depth.copy()
        assert d1.shape == (T, H, W) and np.isfinite(d1).all() and d1.min() >= 1 / 1.1 - 1e-5 and d1.max() <= 10 + 1e-4
        assert d1.std() > 1e-3                      # not a constant map
        r2 = pipe(frames, num_inference_steps=1, window_size=T, noise_latents=nl, noise_aug=na)
        assert np.array_equal(r2.depth, d1)
    finally:
        pipe.engine.close()


def test_device_metrics_match_reference_goldens(engine):
    """G5: the reference's own depth_evaluation(align_with_lstsq=True, custom_mask) / normal_evaluation outputs."""
    res, (s, t_) = engine.eval_depth(G["g5_gt_d"], G["g5_mask"], pred=G["g5_pred_d"])
    for k, v in zip(G["g5_depth_keys"], G["g5_depth_vals"]):
        assert res[str(k)] == pytest.approx(float(v), rel=3e-5, abs=1e-6), k
    nres = engine.eval_normal(G["g5_gt_n"], G["g5_mask"], pred=G["g5_pred_n"])
    for k, v in zip(G["g5_normal_keys"], G["g5_normal_vals"]):
        assert nres[str(k)] == pytest.approx(float(v), rel=3e-5, abs=2e-4), k


def test_device_metrics_large_and_host_mirror(engine):
    """25 x 384 x 512: device metrics == host mirror (exact median selection included)."""
    from unigeo_amd.harness import depth_evaluation, normal_evaluation
    rng = np.random.default_rng(5)
    shape = (25, 384, 512)
    gt = rng.uniform(0.3, 9.0, shape).astype(np.float32); gt[:, :5] = 0
    pred = (1.7 * gt - 0.2 + 0.3 * rng.standard_normal(shape)).astype(np.float32)
    mask = rng.uniform(size=shape) > 0.1
    got, _ = engine.eval_depth(gt, mask, pred=pred)
    ref = depth_evaluation(pred, gt, custom_mask=mask, align_with_lstsq=True)[0]
    for k in ref:
        assert got[k] == pytest.approx(ref[k], rel=2e-4, abs=1e-6), k
    gn = rng.standard_normal(shape + (3,)).astype(np.float32); gn /= np.linalg.norm(gn, axis=-1, keepdims=True)
    pn = (gn + 0.2 * rng.standard_normal(gn.shape)).astype(np.float32)
    gotn = engine.eval_normal(gn, mask, pred=pn)
    refn = normal_evaluation(pn, gn, custom_mask=mask)
    for k in refn:
        assert gotn[k] == pytest.approx(refn[k], rel=1e-4, abs=1e-3), k


def test_harness_with_device_metrics(tiny_plugin, tmp_path):
    from unigeo_amd.harness import SyntheticGeometryDataset, evaluate
    cfg = {"root": "x", "h": 64, "w": 64, "clip_length": 3, "clip_overlap": 1,
           "eval_depth": {"metric_names": ["Abs Rel", "delta < 1.25"]}, "eval_normal": {"metric_names": ["normal mean", "normal median"]}}
    ds = SyntheticGeometryDataset(clip_length=3, clip_overlap=1, input_size=(64, 64), num_frames=5)
    host, _ = evaluate(cfg, dataset=ds, model=tiny_plugin, save_dir=str(tmp_path / "h"), verbose=False)
    dev, _ = evaluate(cfg, dataset=ds, model=tiny_plugin, save_dir=str(tmp_path / "d"), verbose=False, device_metrics=True)
    for a, b in zip(host, dev):
        for k in ("Abs Rel", "delta < 1.25", "normal mean", "normal median"):
            assert b[k] == pytest.approx(a[k], rel=2e-4, abs=1e-3), k


def test_multi_gpu_entry_points_run_under_a_single_rank_rccl_group(tmp_path):
    """BASELINE configs[2] plumbing: the exact code the driver launches on 8 GPUs - bench.py's N > 1 branch (RCCL group, all_gather
    of engine memory, barriers, max-over-ranks timing) and tools/eval_sharded.py (sharded evaluate -> all_gather_object -> one CSV) -
    executed end to end as subprocesses with a forced one-rank nccl group, so the first 8-GPU run cannot die on plumbing."""
    import json, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    env["MASTER_PORT"] = "29561"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--tiny", "--steps", "2", "--warmup", "1",
                        "--frames", "5", "--height", "64", "--width", "128", "--denoise-steps", "2", "--no-cpu-baseline", "--in-flight", "2"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])          # the JSON line must be the LAST line of stdout (driver contract)
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["scaling"] == "weak" and "all_gather" in line["config"]["parallelism"]
    assert line["config"]["clips_in_flight_per_gpu"] == 2      # min(--in-flight, --steps) engine contexts per rank (the throughput mode; the default is 1), the gathers issued in clip order from the main thread
    cfg = tmp_path / "cfg.yaml"
    cfg.write_text(textwrap.dedent("""
        dataset: "SyntheticGeometryDataset"
        root: "unused"
        h: 64
        w: 64
        clip_length: 9
        clip_overlap: 1
        model_name: "DepthCrafter"
        model_params: {synthetic_weights: true, tiny: true, num_inference_steps: 2, workspace_bytes: 3221225472}
        eval_depth: {metric_names: ['Abs Rel', 'delta < 1.25'], depth_alignment: "lstsq"}
        eval_normal: {metric_names: ['normal mean']}
        """))
    env["MASTER_PORT"] = "29562"; env["UG_FORCE_DIST"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "eval_sharded.py"), str(cfg)], capture_output=True, text=True,
                       env=env, cwd=str(tmp_path), timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    csv = (tmp_path / "debug_output" / "metrics.csv").read_text().strip().splitlines()
    assert csv[-1].startswith("Average") and len(csv) >= 3
