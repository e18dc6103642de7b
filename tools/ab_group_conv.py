#!/usr/bin/env python
"""A/B of the im2col K order on the clip's conv shapes: knob 128 = tap-major [tap][channel], 0 = chunk-major [64-channel chunk][tap]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=24 << 30, persist_bytes=64 << 20)
convs = [("unet320@48x64", 320, dict(T=25, H=48, W=64, C0=320, C1=0, kt=1, k=3)),
         ("unet640cat@48x64", 320, dict(T=25, H=48, W=64, C0=320, C1=320, kt=1, k=3)),
         ("unet640@24x32", 640, dict(T=25, H=24, W=32, C0=640, C1=0, kt=1, k=3)),
         ("unet1280cat@24x32", 640, dict(T=25, H=24, W=32, C0=640, C1=640, kt=1, k=3)),
         ("unet1920cat@24x32", 640, dict(T=25, H=24, W=32, C0=1280, C1=640, kt=1, k=3)),
         ("unet1280@12x16", 1280, dict(T=25, H=12, W=16, C0=1280, C1=0, kt=1, k=3)),
         ("unet2560cat@12x16", 1280, dict(T=25, H=12, W=16, C0=1280, C1=1280, kt=1, k=3)),
         ("tconv1280@12x16", 1280, dict(T=25, H=12, W=16, C0=1280, C1=0, kt=3, k=1)),
         ("tconv640@24x32", 640, dict(T=25, H=24, W=32, C0=640, C1=0, kt=3, k=1)),
         ("vae512@96x128", 512, dict(T=8, H=96, W=128, C0=512, C1=0, kt=1, k=3)),
         ("vae512@48x64", 512, dict(T=8, H=48, W=64, C0=512, C1=0, kt=1, k=3)),
         ("vae256@192x256", 256, dict(T=8, H=192, W=256, C0=256, C1=0, kt=1, k=3)),
         ("vae128@384x512", 128, dict(T=8, H=384, W=512, C0=128, C1=0, kt=1, k=3)),
         ("tconv320@48x64", 320, dict(T=25, H=48, W=64, C0=320, C1=0, kt=3, k=1)),
         ("tconv128@384x512", 128, dict(T=8, H=384, W=512, C0=128, C1=0, kt=3, k=1))]
for name, N, cv in convs:
    row = []
    for knob in (128, 0, 128, 0):
        eng.tune_force(-100 - knob, 0)
        ms, tf, c, s = eng.bench_gemm(N=N, conv=cv, iters=20)
        row.append(f"{'tap-major  ' if knob else 'chunk-major'} {ms * 1e3:7.1f} us {tf:5.0f} TF/s")
    print(f"{name:20s} cfg {c:2d} split {s}: " + " | ".join(row), flush=True)
eng.tune_force(-100, 0)
