#!/usr/bin/env python
"""Split-K factor sweep on the lowest-resolution UNet level (M = 25*6*8 = 1200 rows), cfg 0 / 1 / 12 / 3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=8 << 30, persist_bytes=64 << 20)
probs = [("lin 1200x1280x5120", dict(M=1200, N=1280, K=5120)), ("lin 1200x1280x1280", dict(M=1200, N=1280, K=1280)),
         ("lin 1200x3840x1280", dict(M=1200, N=3840, K=1280)),
         ("conv1280@6x8", dict(N=1280, conv=dict(T=25, H=6, W=8, C0=1280, C1=0, kt=1, k=3))),
         ("conv2560cat@6x8", dict(N=1280, conv=dict(T=25, H=6, W=8, C0=1280, C1=1280, kt=1, k=3))),
         ("tconv1280@6x8", dict(N=1280, conv=dict(T=25, H=6, W=8, C0=1280, C1=0, kt=3, k=1))),
         ("conv2560cat@12x16", dict(N=1280, conv=dict(T=25, H=12, W=16, C0=1280, C1=1280, kt=1, k=3))),
         ("conv1280@12x16", dict(N=1280, conv=dict(T=25, H=12, W=16, C0=1280, C1=0, kt=1, k=3)))]
for name, kw in probs:
    row = []
    for cfg in (0, 1, 3, 12):
        for sp in (1, 2, 3, 4, 5, 6, 8):
            best = 0
            for r in range(2):
                try:
                    ms, tf, c, s = eng.bench_gemm(cfg=cfg, split=sp, iters=10, **kw)
                    best = max(best, tf)
                except RuntimeError:
                    pass
            row.append((best, f"c{cfg}/s{sp}"))
    row.sort(reverse=True)
    ms, tf, c, s = eng.bench_gemm(iters=10, **kw)
    print(f"{name:20s} auto c{c}/s{s} {tf:5.0f} | " + "  ".join(f"{k}:{v:4.0f}" for v, k in row[:8]), flush=True)
