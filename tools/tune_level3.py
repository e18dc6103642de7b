#!/usr/bin/env python
"""Tile-config x split-K sweep on the lowest-resolution UNet level (M = 25*6*8 = 1200 rows) INCLUDING the producer / consumer and 3-stage
tiles (tune_splitk.py only tried the symmetric 2-stage ones): these launches stream 10 - 60 MB of weights from HBM through 36 - 72 K steps per
workgroup, so what matters is how far the fetch runs ahead, not the MFMA rate.  Cache-cold (ug_bench_gemm rotates buffers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=8 << 30, persist_bytes=64 << 20)
probs = [("conv1280@6x8", dict(N=1280, conv=dict(T=25, H=6, W=8, C0=1280, C1=0, kt=1, k=3))),
         ("conv2560cat@6x8", dict(N=1280, conv=dict(T=25, H=6, W=8, C0=1280, C1=1280, kt=1, k=3))),
         ("tconv1280@6x8", dict(N=1280, conv=dict(T=25, H=6, W=8, C0=1280, C1=0, kt=3, k=1))),
         ("lin 1200x1280x5120", dict(M=1200, N=1280, K=5120)), ("lin 1200x1280x1280", dict(M=1200, N=1280, K=1280)),
         ("lin 1200x3840x1280", dict(M=1200, N=3840, K=1280))]
for name, kw in probs:
    row = []
    for cfg in (0, 3, 19, 4, 59, 63, 61, 14):
        for sp in (1, 2, 3, 4, 5, 6, 8, 10):
            best = 1e9
            for r in range(2):
                try:
                    ms, tf, c, s = eng.bench_gemm(cfg=cfg, split=sp, iters=10, **kw)
                    best = min(best, ms)
                except RuntimeError:
                    pass
            if best < 1e9:
                row.append((best * 1000, f"c{cfg}/s{sp}"))
    row.sort()
    ms, tf, c, s = eng.bench_gemm(iters=10, **kw)
    print(f"{name:20s} auto c{c}/s{s} {ms * 1000:6.1f} us | " + "  ".join(f"{k}:{v:5.1f}" for v, k in row[:10]), flush=True)
