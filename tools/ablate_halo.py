#!/usr/bin/env python
"""Timing-only ablation (knob 262144: WRONG results): the producer / consumer im2col GEMM with the activation rows fetched for one tap of nine
(the other eight read zeros without touching memory) = the L2 -> LDS traffic a halo-resident tile would have.  Upper bound of what a
halo-staged 3x3 convolution can gain over the per-tap im2col fetch.  Shapes = the 3x3 convolutions of one clip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=24 << 30, persist_bytes=64 << 20)
shapes = [("unet L0 320->320", 320, dict(T=25, H=48, W=64, C0=320, C1=0, kt=1, k=3)),
          ("unet L0 640->320", 320, dict(T=25, H=48, W=64, C0=320, C1=320, kt=1, k=3)),
          ("unet L1 640->640", 640, dict(T=25, H=24, W=32, C0=640, C1=0, kt=1, k=3)),
          ("unet L2 1280->1280", 1280, dict(T=25, H=12, W=16, C0=1280, C1=0, kt=1, k=3)),
          ("vae 128->128 @384x512", 128, dict(T=8, H=384, W=512, C0=128, C1=0, kt=1, k=3)),
          ("vae 256->256 @192x256", 256, dict(T=8, H=192, W=256, C0=256, C1=0, kt=1, k=3)),
          ("vae 512->512 @96x128", 512, dict(T=8, H=96, W=128, C0=512, C1=0, kt=1, k=3))]
print(f"{'shape':26s} {'cfg':>4s} {'full us':>9s} {'TF/s':>7s} | {'1-tap-fetch us':>14s} {'TF/s-equiv':>10s}  ratio")
for name, N, cv in shapes:
    eng.tune_force(-100 - 0, -1)
    full = min(eng.bench_gemm(N=N, conv=cv, iters=20) for _ in range(3))
    eng.tune_force(-100 - 262144, -1)
    abl = min(eng.bench_gemm(N=N, conv=cv, iters=20) for _ in range(3))
    eng.tune_force(-100 - 0, -1)
    print(f"{name:26s} {full[2]:4d} {full[0]*1e3:9.1f} {full[1]:7.0f} | {abl[0]*1e3:14.1f} {abl[1]:10.0f}  {abl[0]/full[0]:.3f}", flush=True)
