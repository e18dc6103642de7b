#!/usr/bin/env python
"""192-row tiles (configs 61 / 62) against the planner's current picks on the M = 19200 / 4800 / 76800 shapes of the clip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=24 << 30, persist_bytes=64 << 20)
dense = [(19200, 640, 2560), (19200, 640, 640), (19200, 1920, 640), (4800, 1280, 5120), (4800, 1280, 1280), (4800, 3840, 1280),
         (76800, 320, 320), (76800, 960, 320), (19200, 5120, 640), (4800, 10240, 1280)]
convs = [("unet640@24x32", 640, dict(T=25, H=24, W=32, C0=640, C1=0, kt=1, k=3)),
         ("unet1280cat@24x32", 640, dict(T=25, H=24, W=32, C0=640, C1=640, kt=1, k=3)),
         ("unet1920cat@24x32", 640, dict(T=25, H=24, W=32, C0=1280, C1=640, kt=1, k=3)),
         ("unet1280@12x16", 1280, dict(T=25, H=12, W=16, C0=1280, C1=0, kt=1, k=3)),
         ("unet2560cat@12x16", 1280, dict(T=25, H=12, W=16, C0=1280, C1=1280, kt=1, k=3)),
         ("unet320@48x64", 320, dict(T=25, H=48, W=64, C0=320, C1=0, kt=1, k=3)),
         ("unet640cat@48x64", 320, dict(T=25, H=48, W=64, C0=320, C1=320, kt=1, k=3)),
         ("tconv640@24x32", 640, dict(T=25, H=24, W=32, C0=640, C1=0, kt=3, k=1)),
         ("tconv1280@12x16", 1280, dict(T=25, H=12, W=16, C0=1280, C1=0, kt=3, k=1)),
         ("tconv320@48x64", 320, dict(T=25, H=48, W=64, C0=320, C1=0, kt=3, k=1))]
def row(name, **kw):
    r = []
    for c in (-1, 63, 64, 61, 59, 35, 15):
        best = 1e9; cc = c
        for _ in range(3):
            a = dict(kw)
            if c >= 0: a.update(cfg=c, split=1)
            ms, tf, cc, ss = eng.bench_gemm(iters=10, **a)
            best = min(best, ms)
        r.append(f"{'auto=c%d/s%d' % (cc, ss) if c < 0 else 'c%d' % c}: {best * 1e3:7.1f}")
    print(f"{name:22s} " + "  ".join(r), flush=True)
for (M, N, K) in dense: row(f"{M}x{N}x{K}", M=M, N=N, K=K)
for (name, N, cv) in convs: row(name, N=N, conv=cv)
