"""MX-fp8 linear layers (BASELINE configs[4]: "DepthCrafter fp8 MFMA ... (CDNA4 fp8)").  The reference has no fp8 path; what is
checked here is (1) that the device quantiser implements OCP MX (e8m0 block scale per 32 K elements, e4m3 RNE elements) bit for bit,
(2) that the MX GEMM equals the exact product of the dequantised operands up to the fp16 output rounding, and (3) how far the fp8
option moves the UNet / the 50-frame 576x768 clip away from the fp16 path and the fp32 oracle - reported, with honest bounds (fp8
does NOT meet north_star's 1e-3; that tolerance belongs to the fp16 path)."""
import numpy as np
import pytest
import torch

from util import assert_close, h16, mx8_quantise, rel_err, report
from oracle_build import oracle_unet

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K,N", [(300, 256, 192), (2500, 640, 1280), (4096, 1280, 640)])
def test_mx8_quantiser_and_gemm(engine, M, K, N):
    rng = np.random.default_rng(M)
    A = h16(rng.standard_normal((M, K)) * np.exp(rng.normal(0, 1.5, (M, 1))))          # rows of very different scale
    A[0, :40] = 0.0                                                                    # an all-zero block
    A[1, 5] = 300.0                                                                    # an outlier that sets its block's scale
    W = h16(rng.standard_normal((N, K)) / np.sqrt(K))
    b = h16(rng.standard_normal(N) * 0.1)
    got, a8, sa = engine.op_linear_mx8(A, W, bias=b, return_quant=True)
    dqa, bytes_ref, e_ref = mx8_quantise(A)
    dqw, _, _ = mx8_quantise(W)
    np.testing.assert_array_equal(a8, bytes_ref)                                       # OCP e4m3, round to nearest even, per-block scale
    e_dev = np.stack([(sa[:, :M] >> (8 * j)) & 0xff for j in range(4)], -1).transpose(1, 0, 2).reshape(M, K // 32)
    np.testing.assert_array_equal(e_dev.astype(np.uint8), e_ref)
    ref = dqa.astype(np.float64) @ dqw.astype(np.float64).T + b
    assert_close(got, ref, 9e-4, f"MX-fp8 GEMM vs exact product of the dequantised operands {M}x{N}x{K}")
    report(f"MX-fp8 linear {M}x{N}x{K}: quantisation error vs the fp16-operand product", rel_err(got, A.astype(np.float64) @ W.T.astype(np.float64) + b))


@pytest.mark.parametrize("pick", [0, 1, 2, 3])
def test_mx8_tile_shapes_bitwise_identical(engine, pick):
    """The four MX tile shapes (256x256, 256x128, 128x128, 192x128) walk K through the same scaled-MFMA chain: outputs must be bit-identical
    (a ragged M, so the 192-row tile's masked scale fetch and edge rows are exercised)."""
    rng = np.random.default_rng(77)
    M, K, N = 2500, 640, 1280
    A, W, b = h16(rng.standard_normal((M, K))), h16(rng.standard_normal((N, K)) / np.sqrt(K)), h16(rng.standard_normal(N) * 0.1)
    try:
        engine.tune_force(101, -1)
        ref = engine.op_linear_mx8(A, W, bias=b)
        engine.tune_force(100 + pick, -1)
        got = engine.op_linear_mx8(A, W, bias=b)
    finally:
        engine.tune_force(-1, -1)
    assert np.array_equal(got, ref), f"MX tile pick {pick}: max diff {np.abs(got - ref).max()}"


def test_mx8_geglu_epilogue(engine):
    rng = np.random.default_rng(9)
    M, K, N = 1024, 640, 512
    A, W, b = h16(rng.standard_normal((M, K))), h16(rng.standard_normal((N, K)) / np.sqrt(K)), h16(rng.standard_normal(N) * 0.1)
    got = engine.op_linear_mx8(A, W, bias=b, geglu=True)
    y = mx8_quantise(A)[0].astype(np.float64) @ mx8_quantise(W)[0].astype(np.float64).T + b
    h, g = torch.from_numpy(y[:, : N // 2]), torch.from_numpy(y[:, N // 2:])
    assert_close(got, (h * torch.nn.functional.gelu(g)).numpy(), 7e-4, "MX-fp8 GEMM with the GEGLU epilogue")


def test_fp8_unet_error_vs_fp16_path_and_oracle():
    """Full architecture, 8 frames of 32x32 latents (2048 / 512 token rows on levels 1 / 2, 8192 on level 0): the UNet output with the
    fp8 linears against the fp16 path and the fp32 oracle."""
    from unigeo_amd import weights as W
    from unigeo_amd.pipeline import DepthCrafterPipelineHIP
    u, v, c = W.UNetCfg(), W.tiny_cfgs()[1], W.tiny_cfgs()[2]
    su, sv, sc = (W.random_state(W.unet_manifest(u), 11), W.random_state(W.vae_manifest(v), 12), W.random_state(W.clip_manifest(c), 13))
    pipe = DepthCrafterPipelineHIP.from_state(su, sv, sc, cfgs=(u, v, c), workspace_bytes=8 << 30, persist_bytes=8 << 30)
    try:
        rng = np.random.default_rng(5)
        T, h, w = 8, 32, 32
        x = h16(rng.standard_normal((T, u.in_channels, h, w)))
        emb = h16(rng.standard_normal((T, u.cross_attention_dim)))
        ts = 0.25 * np.log(11.0)
        eng = pipe.engine
        y16 = eng.unet_forward(x, ts, emb)
        eng.set_fp8_linears(True)
        y8 = eng.unet_forward(x, ts, emb)
        y8b = eng.unet_forward(x, ts, emb)
        import os
        os.environ["UG_NO_LNQ"] = "1"                     # LayerNorm -> fp16 -> separate quantiser pass instead of the fused MX-fp8 output
        try:
            y8c = eng.unet_forward(x, ts, emb)
        finally:
            del os.environ["UG_NO_LNQ"]
        assert np.array_equal(y8, y8c)                    # the fused quantisation is bit-identical to the two-pass one
        eng.set_fp8_linears(False)
        assert np.array_equal(y8, y8b) and not np.array_equal(y8, y16)                 # deterministic, and really a different path
        unet = oracle_unet(u, su)
        with torch.no_grad():
            ref = unet(torch.from_numpy(x)[None], torch.tensor(ts), torch.from_numpy(emb)[None], torch.tensor([[7.0, 127.0, 0.02]]))[0].numpy()
        e16 = report("UNet (full architecture, 8x32x32): fp16 path vs oracle", rel_err(y16, ref))
        e8 = report("UNet (full architecture, 8x32x32): fp8-linear path vs oracle", rel_err(y8, ref))
        report("UNet (full architecture, 8x32x32): fp8-linear path vs fp16 path", rel_err(y8, y16))
        report("UNet (full architecture, 8x32x32): fp8-linear path vs oracle, rms / rms", float(np.sqrt(((y8 - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean())))
        assert np.isfinite(y8).all() and e16 < 4e-3 and e8 < 1.2e-1, (e16, e8)      # measured 1.9e-3 / 6.2e-2
    finally:
        pipe.engine.close()


def test_fp8_twin_of_larger_clip_geometry():
    """BASELINE configs[4] geometry (50 frames at 576 x 768, 1 Euler step) with the fp8 linears on: finite, in range, reproducible, and
    within a stated distance of the fp16 path's decoded frames."""
    from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
    from unigeo_amd.synthetic import synthetic_clip
    from unigeo_amd.model.depthcrafter import DepthCrafter
    pipe = DepthCrafterPipelineHIP.from_random(seed=42, workspace_bytes=64 << 30)
    try:
        T, H, W = 50, 576, 768
        frames = DepthCrafter.prepare_input(None, synthetic_clip(T, H, W))
        nl, na = make_noise(T, H, W, 0)
        f16_frames = pipe(frames, num_inference_steps=1, window_size=T, noise_latents=nl, noise_aug=na).frames[0].copy()
        pipe.engine.set_fp8_linears(True)
        r1 = pipe(frames, num_inference_steps=1, window_size=T, noise_latents=nl, noise_aug=na)
        f8, d8 = r1.frames[0].copy(), r1.depth.copy()
        r2 = pipe(frames, num_inference_steps=1, window_size=T, noise_latents=nl, noise_aug=na)
        assert np.array_equal(r2.frames[0], f8)
        assert np.isfinite(f8).all() and f8.min() >= 0 and f8.max() <= 1 and d8.min() >= 1 / 1.1 - 1e-5 and d8.max() <= 10 + 1e-4
        emax = report("50x576x768, 1 step: fp8-linear frames vs fp16-path frames, max abs", np.abs(f8 - f16_frames).max())
        emean = report("50x576x768, 1 step: fp8-linear frames vs fp16-path frames, mean abs", np.abs(f8 - f16_frames).mean())
        assert emean < 2e-2 and emax < 0.27, (emean, emax)                             # measured 9.6e-3 / 0.13
    finally:
        pipe.engine.close()
