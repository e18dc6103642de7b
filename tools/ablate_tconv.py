#!/usr/bin/env python
"""Timing-only ablation (knob 262144: WRONG results) for the temporal (3,1,1) convolutions: the producer / consumer im2col GEMM with the activation rows
of ONE tap of three fetched (the other two read zeros without touching memory) - an upper bound of what a tile that stages a (frames + 2) halo once
for its three taps could gain (such a tile would still fetch 7/5 of one tap)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=24 << 30, persist_bytes=64 << 20)
shapes = [("unet L0 320", 320, dict(T=25, H=48, W=64, C0=320, C1=0, kt=3, k=1)),
          ("unet L1 640", 640, dict(T=25, H=24, W=32, C0=640, C1=0, kt=3, k=1)),
          ("unet L2 1280", 1280, dict(T=25, H=12, W=16, C0=1280, C1=0, kt=3, k=1)),
          ("vae 128 @384x512", 128, dict(T=8, H=384, W=512, C0=128, C1=0, kt=3, k=1)),
          ("vae 256 @192x256", 256, dict(T=8, H=192, W=256, C0=256, C1=0, kt=3, k=1)),
          ("vae 512 @96x128", 512, dict(T=8, H=96, W=128, C0=512, C1=0, kt=3, k=1))]
print(f"{'shape':20s} {'planner cfg/us':>16s} | {'cfg':>4s} {'full us':>9s} {'1-tap-fetch us':>14s}  ratio")
for name, N, cv in shapes:
    eng.tune_force(-100 - 0, -1)
    pl = min(eng.bench_gemm(N=N, conv=cv, iters=20) for _ in range(3))
    for cfg in (63, 64, 59, 54):
        eng.tune_force(-100 - 0, -1)
        full = min(eng.bench_gemm(N=N, conv=cv, cfg=cfg, split=1, iters=20) for _ in range(3))
        eng.tune_force(-100 - 262144, -1)
        abl = min(eng.bench_gemm(N=N, conv=cv, cfg=cfg, split=1, iters=20) for _ in range(3))
        eng.tune_force(-100 - 0, -1)
        print(f"{name:20s} {pl[2]:6d} {pl[0]*1e3:9.1f} | {cfg:4d} {full[0]*1e3:9.1f} {abl[0]*1e3:14.1f}  {abl[0]/full[0]:.3f}", flush=True)
