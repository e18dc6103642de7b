#!/usr/bin/env python
"""Halo-staged 3x3 convolution (kernels/conv_halo.hip, default on; knob 16384 = off) against the im2col GEMM: bit-identity on small problems, then us per
launch on the 3x3 convolutions of one clip (planner's im2col choice vs halo).  usage: ab_halo.py [check|time|all]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
what = sys.argv[1] if len(sys.argv) > 1 else "all"
eng = Engine(0, workspace_bytes=24 << 30, persist_bytes=64 << 20)
OFF, L0 = 16384, 0              # knobs: 16384 = halo kernel off (im2col everywhere); default = halo wherever it is stageable (32768 = level 0 back on the row-split im2col pair)
if what in ("check", "all"):
    rng = np.random.default_rng(0)
    ok = True
    for (T, H, W, C0, C1, O) in [(1, 16, 16, 64, 0, 128), (2, 48, 64, 128, 0, 320), (2, 24, 32, 128, 64, 160), (2, 32, 48, 64, 0, 128),
                                 (1, 64, 128, 128, 0, 128), (3, 48, 64, 64, 64, 320), (25, 48, 64, 64, 0, 320), (1, 32, 32, 192, 0, 256), (2, 16, 256, 64, 0, 96),
                                 (2, 12, 16, 128, 0, 256), (4, 24, 32, 64, 64, 640), (25, 24, 32, 64, 0, 640), (3, 12, 16, 128, 128, 160)]:
        x0 = rng.standard_normal((T, H, W, C0)).astype(np.float32)
        x1 = rng.standard_normal((T, H, W, C1)).astype(np.float32) if C1 else None
        w = (rng.standard_normal((O, C0 + C1, 3, 3)) * (9 * (C0 + C1)) ** -0.5).astype(np.float32)
        b = rng.standard_normal(O).astype(np.float32)
        eng.tune_force(-100 - OFF, -1)
        ref = eng.op_conv(x0, w, b, x1=x1)
        eng.tune_force(-100 - L0, -1)
        got = eng.op_conv(x0, w, b, x1=x1)
        eng.tune_force(-100 - 0, -1)
        same = np.array_equal(ref, got)
        ok &= same
        print(f"check T{T} {H}x{W} C{C0}+{C1} -> {O}: {'bit-identical' if same else 'MISMATCH max |d| = %.4g (ref max %.3g)' % (np.abs(ref - got).max(), np.abs(ref).max())}", flush=True)
    print("CHECK", "OK" if ok else "FAILED", flush=True)
if what in ("time", "all"):
    shapes = [("unet L0 320->320", 320, dict(T=25, H=48, W=64, C0=320, C1=0, kt=1, k=3)),
              ("unet L0 640->320", 320, dict(T=25, H=48, W=64, C0=320, C1=320, kt=1, k=3)),
              ("unet L0 960->320", 320, dict(T=25, H=48, W=64, C0=640, C1=320, kt=1, k=3)),
              ("unet L1 640->640", 640, dict(T=25, H=24, W=32, C0=640, C1=0, kt=1, k=3)),
              ("unet L1 1280->640", 640, dict(T=25, H=24, W=32, C0=640, C1=640, kt=1, k=3)),
              ("unet L1 320->640", 640, dict(T=25, H=24, W=32, C0=320, C1=0, kt=1, k=3)),
              ("unet L2 1280->1280", 1280, dict(T=25, H=12, W=16, C0=1280, C1=0, kt=1, k=3)),
              ("unet L2 2560->1280", 1280, dict(T=25, H=12, W=16, C0=1280, C1=1280, kt=1, k=3)),
              ("vae 128->128 @384x512", 128, dict(T=8, H=384, W=512, C0=128, C1=0, kt=1, k=3)),
              ("vae 256->128 @384x512", 128, dict(T=8, H=384, W=512, C0=256, C1=0, kt=1, k=3)),
              ("vae 256->256 @192x256", 256, dict(T=8, H=192, W=256, C0=256, C1=0, kt=1, k=3)),
              ("vae 512->256 @192x256", 256, dict(T=8, H=192, W=256, C0=512, C1=0, kt=1, k=3)),
              ("vae 512->512 @96x128", 512, dict(T=8, H=96, W=128, C0=512, C1=0, kt=1, k=3)),
              ("vae 512->512 @48x64", 512, dict(T=8, H=48, W=64, C0=512, C1=0, kt=1, k=3))]
    print(f"{'shape':24s} {'cfg':>4s} {'im2col us':>10s} {'TF/s':>7s} | {'halo us':>9s} {'TF/s':>7s}  ratio")
    for name, N, cv in shapes:
        eng.tune_force(-100 - OFF, -1)
        a = min(eng.bench_gemm(N=N, conv=cv, iters=20) for _ in range(3))
        eng.tune_force(-100 - L0, -1)
        b = min(eng.bench_gemm(N=N, conv=cv, iters=20) for _ in range(3))
        eng.tune_force(-100 - 0, -1)
        print(f"{name:24s} {a[2]:4d} {a[0]*1e3:10.1f} {a[1]:7.0f} | {b[0]*1e3:9.1f} {b[1]:7.0f}  {b[0]/a[0]:.3f}", flush=True)
