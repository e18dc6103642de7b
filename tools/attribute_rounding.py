#!/usr/bin/env python
"""Round 5 (VERDICT r4 "Next" 4): which of the HIP engine's fp16 rounding points account for its distance from the fp32 oracle?  The tiny full-topology
configuration over the 25-step trajectory of tests/test_trajectory_gpu.py::test_tiny_25_step_trajectory_fp16_vs_fp16, with one switch at a time moved away from
the default.  Prints max over steps of max |latent error| / max |latent| against the fp32 oracle and against the fp16-storage run of the oracle.
usage (GPU box): python tools/attribute_rounding.py"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.pipeline import run_pipeline
from oracle_build import oracle_clip, oracle_unet, oracle_vae
from unigeo_amd import weights as W
from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
from util import fp16_storage

u, v, c = W.tiny_cfgs()
su, sv, sc = (W.random_state(W.unet_manifest(u), 1), W.random_state(W.vae_manifest(v), 2), W.random_state(W.clip_manifest(c), 3))
T, H, Wd, steps, seed = 5, 64, 64, 25, 21
rng = np.random.default_rng(seed)
yy, xx = np.mgrid[0:H, 0:Wd].astype(np.float32)
base = 127.5 + 90.0 * np.sin(2 * np.pi * (xx / 37.0 + yy / 29.0))[None, :, :, None]
frames = np.clip(base + 25.0 * rng.standard_normal((T, H, Wd, 3)) + 6.0 * np.arange(T)[:, None, None, None], 0, 255).astype(np.uint8).astype(np.float32) / 255.0
nl, na = make_noise(T, H, Wd, seed=seed)
ou, ov, oc = oracle_unet(u, su), oracle_vae(v, sv), oracle_clip(c, sc)
_, st32 = run_pipeline(ou, ov, oc, frames, torch.from_numpy(nl), torch.from_numpy(na), steps=steps, return_stages=True)
with fp16_storage(ou, ov.decoder, oc):
    _, st16 = run_pipeline(ou, ov, oc, frames, torch.from_numpy(nl), torch.from_numpy(na), steps=steps, return_stages=True)
r32 = np.stack([x.numpy() for x in st32["latents_per_step"]], 0).astype(np.float64)
r16 = np.stack([x.numpy() for x in st16["latents_per_step"]], 0).astype(np.float64)
scl = np.abs(r32).max(axis=(1, 2, 3, 4))
d = lambda a, b: (np.abs(a - b).max(axis=(1, 2, 3, 4)) / scl)
pipe = DepthCrafterPipelineHIP.from_state(su, sv, sc, cfgs=(u, v, c), workspace_bytes=3 << 30)
eng = pipe.engine
VARIANTS = [("default", lambda: None, lambda: None),
            ("feed-forward as two GEMM launches (fp16 GEGLU intermediate through HBM)", lambda: eng.set_ff_fused(False), lambda: eng.set_ff_fused(True)),
            ("flash attention: reference maximum tracked per tile (variant 7: no lazy rescale / fp32 row sums)", lambda: eng.tune_flash(7), lambda: eng.tune_flash(-1)),
            ("VAE encoder in fp16 storage instead of float32-grade", lambda: eng.set_vae_encode_fp32(False), lambda: eng.set_vae_encode_fp32(True)),
            ("GEMM tiles forced to 128 x 128 (different split / summation order)", lambda: eng.tune_force(0, 1), lambda: eng.tune_force(-1, -1))]
print(f"fp16-storage run of the oracle vs fp32 oracle: {d(r16, r32).max():.2e} (max over the 25 steps)")
print(f"{'variant':100s}  HIP-vs-fp32  HIP-vs-fp16run  step-1  step-25 (vs fp32)")
try:
    eng.set_inputs(frames, nl, na, None)
    for name, on, off in VARIANTS:
        on()
        try:
            hip = eng.run_traced(steps, 8, with_normals=False).astype(np.float64)
        finally:
            off()
        e32, e16 = d(hip, r32), d(hip, r16)
        print(f"{name:100s}  {e32.max():.2e}     {e16.max():.2e}        {e32[0]:.2e}  {e32[-1]:.2e}", flush=True)
finally:
    eng.close()
