"""StableNormal (BASELINE configs[3]; reference model/stablenormal.py:16,39) through the C ABI against the CPU oracle's restatement
(oracle/stablenormal.py - UNPINNED): DINO tower, SD VAE, the two UNets with and without ControlNet residuals
(77-token text cross-attention), the whole predictor on a tiny full-topology configuration, then the REAL architecture
(865.9 M-parameter UNets, 363 M ControlNets, ViT-L/14, SD VAE; seeded random weights) incl. one 576x576 image end to end.
Tolerances are <= ~2x the values measured on MI355X (profiles/r02_parity_measured.jsonl)."""
import numpy as np
import pytest
import torch

from util import assert_abs, assert_close, h16, report
from oracle_build import oracle_stablenormal

pytestmark = pytest.mark.gpu


def _angle(a, b):
    return np.degrees(np.arccos(np.clip((a * b).sum(-1), -1, 1)))


def _build(cfgs, seed, **kw):
    from unigeo_amd import weights as W
    from unigeo_amd.stablenormal import COMPONENTS, StableNormalPredictorHIP, manifests
    ms = manifests(cfgs)
    states = {c: W.random_state(ms[c], seed + i) for i, c in enumerate(COMPONENTS)}
    pe = h16(np.random.default_rng(seed + 100).standard_normal((77, cfgs[0].cross_attention_dim)))
    pred = StableNormalPredictorHIP.from_states(states, cfgs, prompt_embeds=pe, **kw)
    return pred, states, pe


@pytest.fixture(scope="module")
def tiny():
    from unigeo_amd import weights as W
    cfgs = W.tiny_sn_cfgs()
    pred, states, pe = _build(cfgs, 50, workspace_bytes=3 << 30, refine_steps=3)
    yield dict(pred=pred, eng=pred.engine, cfgs=cfgs, pe=pe, **oracle_stablenormal(cfgs, states))
    pred.engine.close()


def test_dino_tokens(tiny):
    from oracle.stablenormal import dino_preprocess
    rng = np.random.default_rng(0)
    img = h16(rng.uniform(0, 1, (2, 64, 128, 3)))
    got = tiny["eng"].sn_dino(img)
    with torch.no_grad():
        ref = tiny["dino"](dino_preprocess(torch.from_numpy(img).permute(0, 3, 1, 2) * 2 - 1, 224)).numpy()
    assert_close(got, ref, 2e-3, "SN tiny DINO patch tokens")


def test_sd_vae(tiny):
    rng = np.random.default_rng(1)
    img = h16(rng.uniform(-1, 1, (2, 64, 64, 3)))
    with torch.no_grad():
        ref_e = tiny["vae"].encode_mode(torch.from_numpy(img).permute(0, 3, 1, 2)).numpy()
    assert_close(tiny["eng"].sn_vae_encode(img), ref_e, 3.5e-3, "SN tiny VAE encode (fp16 storage)")
    z = h16(rng.standard_normal((2, 4, 8, 16)) * 2)
    with torch.no_grad():
        ref_d = tiny["vae"].decode(torch.from_numpy(z)).permute(0, 2, 3, 1).numpy()
    assert_close(tiny["eng"].sn_vae_decode(z), ref_d, 4e-3, "SN tiny VAE 2-D decode")


@pytest.mark.parametrize("which,use_ctrl", [(0, False), (0, True), (1, True)])
def test_unet_with_controlnet(tiny, which, use_ctrl):
    rng = np.random.default_rng(10 + which)
    B, h, w = 2, 8, 16
    u, _, d = tiny["cfgs"]
    x, z = h16(rng.standard_normal((B, 4, h, w))), h16(rng.standard_normal((B, 4, h, w)))
    tok = h16(rng.standard_normal((B, 256, d.hidden_size))) if which == 1 else None
    t_u, t_c = (999.0, 999.0) if which == 0 else (321.0, 0.0)
    got = tiny["eng"].sn_unet_forward(which, x, t_u, tiny["pe"], zimg=z, t_ctrl=t_c, dino_tokens=tok, use_ctrl=use_ctrl)
    ctx = torch.from_numpy(tiny["pe"])[None].expand(B, -1, -1)
    unet, ctrl = (tiny["unet_y"], tiny["ctrl_y"]) if which == 0 else (tiny["unet_r"], tiny["ctrl_d"])
    with torch.no_grad():
        dr = mr = None
        if use_ctrl:
            dr, mr = ctrl(torch.from_numpy(z), t_c, ctx, **({"dino_tokens": torch.from_numpy(tok)} if which == 1 else {}))
        ref = unet(torch.from_numpy(x), t_u, ctx, dr, mr).numpy()
    assert_close(got, ref, 4e-3, f"SN tiny UNet which={which} ctrl={use_ctrl}")


@pytest.mark.parametrize("B,H,W", [(1, 64, 64), (3, 64, 128)])
def test_predictor_end_to_end(tiny, B, H, W):
    from oracle.stablenormal import run_stablenormal
    rng = np.random.default_rng(B)
    img = (rng.uniform(0, 255, (B, H, W, 3)).astype(np.uint8)).astype(np.float32) / 255.0
    got = tiny["pred"].predict_batch(img)
    ref = run_stablenormal(tiny["vae"], tiny["unet_y"], tiny["ctrl_y"], tiny["unet_r"], tiny["ctrl_d"], tiny["dino"], img, tiny["pe"],
                           refine_steps=3)
    assert got.shape == ref.shape == (B, H, W, 3) and np.isfinite(got).all()
    np.testing.assert_allclose(np.linalg.norm(got, axis=-1), 1.0, atol=1e-4)
    ang = _angle(got, ref)
    report(f"SN tiny predictor B={B} {H}x{W}: mean angle to oracle (deg)", ang.mean()); report(f"SN tiny predictor B={B} {H}x{W}: max angle (deg)", ang.max())
    assert ang.mean() < 0.5 and np.percentile(ang, 99) < 3.0, (ang.mean(), np.percentile(ang, 99))
    # frames are independent (spatial-only network): a batch equals its frames run one by one
    if B > 1:
        one = tiny["pred"].predict_batch(img[1:2])
        assert _angle(one[0], got[1]).max() < 0.2


def test_plugin_contract_and_uint8_postprocessing(tiny):
    """forward(data) -> {'pred_normals' [Nf,H,W,3] in [-1,1], 'pred_depths' zeros} (reference model/stablenormal.py:30-52)."""
    from unigeo_amd.model import StableNormal
    from unigeo_amd.stablenormal import normals_to_uint8
    from unigeo_amd.synthetic import synthetic_clip
    m = StableNormal(predictor=tiny["pred"])
    data = synthetic_clip(2, 64, 64, seed=5)
    out = m.forward(data)
    n, d = out["pred_normals"], out["pred_depths"]
    assert n.shape == (2, 64, 64, 3) and d.shape == (2, 64, 64) and n.dtype == torch.float32 and float(d.abs().max()) == 0.0
    raw = tiny["pred"].predict_batch(m.prepare_input(data))
    u8 = normals_to_uint8(raw).copy()
    u8[..., 0] = -u8[..., 0]                                          # the reference's uint8 wrap (:43)
    np.testing.assert_array_equal(n.numpy(), (u8 / 255.0 * 2 - 1).astype(np.float32))
    with pytest.raises(FileNotFoundError):
        StableNormal(model_dir="/nonexistent")                        # no silent fallback
    with pytest.raises(ValueError):
        tiny["pred"].predict_batch(np.zeros((1, 60, 64, 3), np.float32))


def test_full_architecture_unet_controlnet_and_576_image():
    """The real sizes: 865.9 M-parameter UNet2DConditionModel x 2, 363 M ControlNet x 2, DINOv2 ViT-L/14, SD VAE (seeded random
    weights).  One UNet + DINO-ControlNet evaluation on a 16x16 latent, then one 576x576 image through the whole predictor
    (BASELINE configs[3] geometry: 72x72 latents, S = 5184 self-attention, 77-token cross-attention) with 2 refinement steps."""
    from oracle.stablenormal import run_stablenormal
    from unigeo_amd import weights as W
    cfgs = (W.SDUNetCfg(), W.VAECfg(), W.DinoCfg())
    pred, states, pe = _build(cfgs, 70, workspace_bytes=12 << 30, refine_steps=2)
    try:
        o = oracle_stablenormal(cfgs, states)
        rng = np.random.default_rng(3)
        B, h, w = 1, 16, 16
        x, z = h16(rng.standard_normal((B, 4, h, w))), h16(rng.standard_normal((B, 4, h, w)))
        tok = h16(rng.standard_normal((B, 256, 1024)))
        got = pred.engine.sn_unet_forward(1, x, 281.0, pe, zimg=z, t_ctrl=0.0, dino_tokens=tok, use_ctrl=True)
        ctx = torch.from_numpy(pe)[None]
        with torch.no_grad():
            dr, mr = o["ctrl_d"](torch.from_numpy(z), 0.0, ctx, dino_tokens=torch.from_numpy(tok))
            ref = o["unet_r"](torch.from_numpy(x), 281.0, ctx, dr, mr).numpy()
        assert_close(got, ref, 4.5e-3, "SN full-architecture UNet + DINO ControlNet")
        yy, xx = np.mgrid[0:576, 0:576].astype(np.float32)
        img = np.stack([127.5 + 100 * np.sin(xx / 41.0 + c) * np.cos(yy / 29.0) for c in range(3)], -1) + rng.normal(0, 8, (576, 576, 3))
        img = (np.clip(img, 0, 255).astype(np.uint8).astype(np.float32) / 255.0)[None]
        gotn = pred.predict_batch(img)
        refn = run_stablenormal(o["vae"], o["unet_y"], o["ctrl_y"], o["unet_r"], o["ctrl_d"], o["dino"], img, pe, refine_steps=2)
        ang = _angle(gotn, refn)
        report("SN full-architecture 576x576 predictor: mean angle to oracle (deg)", ang.mean())
        report("SN full-architecture 576x576 predictor: 99th percentile angle (deg)", np.percentile(ang, 99))
        assert np.isfinite(gotn).all() and ang.mean() < 0.5 and np.percentile(ang, 99) < 3.0, (ang.mean(), np.percentile(ang, 99))
    finally:
        pred.engine.close()


def test_device_resize_matches_torch_antialias_bilinear(tiny):
    """ug_resize_bilinear == torch F.interpolate(mode="bilinear", align_corners=False, antialias=True), down and up, ragged sizes."""
    import torch.nn.functional as F
    eng = tiny["pred"].engine
    rng = np.random.default_rng(3)
    for (Hi, Wi, Ho, Wo) in [(96, 128, 64, 64), (64, 64, 96, 160), (100, 75, 37, 41), (48, 64, 48, 64), (33, 47, 128, 192)]:
        x = rng.uniform(-1, 1, (2, Hi, Wi, 3)).astype(np.float32)
        ref = F.interpolate(torch.from_numpy(x).permute(0, 3, 1, 2), size=(Ho, Wo), mode="bilinear", align_corners=False, antialias=True).permute(0, 2, 3, 1).numpy()
        got = eng.resize_bilinear(x, Ho, Wo)
        assert got.shape == ref.shape
        e = float(np.abs(got - ref).max())
        report(f"resize_bilinear {Hi}x{Wi}->{Ho}x{Wo} max abs err", e)
        assert e < 2e-5, (Hi, Wi, Ho, Wo, e)
    n = eng.resize_bilinear(rng.uniform(-1, 1, (1, 40, 40, 3)).astype(np.float32), 64, 64, normalise=True)
    assert np.allclose(np.linalg.norm(n, axis=-1), 1.0, atol=1e-5)


def test_processing_resolution_knob(tiny):
    """processing_resolution = R: resize in (longer side R, multiples of 64) -> predict -> resize back + re-normalise; 0 / R == input size: identity
    path.  Through the YAML-facing plugin kwargs as well."""
    from unigeo_amd.model import StableNormal
    pred = tiny["pred"]
    rng = np.random.default_rng(8)
    img = rng.uniform(0, 1, (1, 128, 192, 3)).astype(np.float32)
    base = pred.predict_batch(img)
    try:
        pred.processing_resolution = 192                       # longer side already 192: same computation
        assert np.array_equal(pred.predict_batch(img), base)
        pred.processing_resolution = 128                       # 128x192 -> 64x128 (rounded to 64s) -> back
        small = pred.predict_batch(img)
        assert small.shape == base.shape and np.isfinite(small).all()
        assert np.allclose(np.linalg.norm(small, axis=-1), 1.0, atol=1e-4)
        want = pred.engine.resize_bilinear(pred.engine.sn_run(pred.engine.resize_bilinear(img, 64, 128), pred.prompt_embeds, pred.yoso_timestep, pred.timesteps, pred.ca, pred.cb),
                                           128, 192, normalise=True)
        assert np.array_equal(small, want)
        ragged = rng.uniform(0, 1, (1, 100, 150, 3)).astype(np.float32)        # not a multiple of 64: only legal with a processing resolution
        assert pred.predict_batch(ragged).shape == (1, 100, 150, 3)
    finally:
        pred.processing_resolution = 0
    with pytest.raises(ValueError):
        pred.predict_batch(rng.uniform(0, 1, (1, 100, 150, 3)).astype(np.float32))


def test_guidance_branch_on_a_second_stream_is_bit_identical(tiny):
    """ug_set_concurrency(2): the DINOv2 tower + DINO ControlNet are issued on a second HIP stream beside the YOSO estimate and joined before the
    refinement loop (the first call of a shape runs in order and measures the branch's arena need).  Same kernels, same launch parameters."""
    pred = tiny["pred"]
    rng = np.random.default_rng(21)
    img = rng.uniform(0, 1, (2, 64, 128, 3)).astype(np.float32)
    pred.engine.set_concurrency(1)
    base = pred.predict_batch(img)
    pred.engine.set_concurrency(2)
    try:
        first = pred.predict_batch(img)        # measures (in order)
        second = pred.predict_batch(img)       # overlapped
        third = pred.predict_batch(img)
    finally:
        pred.engine.set_concurrency(2)
    assert np.array_equal(first, base) and np.array_equal(second, base) and np.array_equal(third, base)
