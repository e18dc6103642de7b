#!/usr/bin/env python
"""Split-K sweep on the tiny-M shapes of a batch-1 StableNormal image (576x576: 81 / 324 / 1296 rows on UNet levels 3 / 2 / 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=8 << 30, persist_bytes=64 << 20)
probs = [("conv1280@9x9", dict(N=1280, conv=dict(T=1, H=9, W=9, C0=1280, C1=0, kt=1, k=3))),
         ("conv2560cat@9x9", dict(N=1280, conv=dict(T=1, H=9, W=9, C0=1280, C1=1280, kt=1, k=3))),
         ("conv1280@18x18", dict(N=1280, conv=dict(T=1, H=18, W=18, C0=1280, C1=0, kt=1, k=3))),
         ("conv2560cat@18x18", dict(N=1280, conv=dict(T=1, H=18, W=18, C0=1280, C1=1280, kt=1, k=3))),
         ("conv640@36x36", dict(N=640, conv=dict(T=1, H=36, W=36, C0=640, C1=0, kt=1, k=3))),
         ("lin 81x1280x1280", dict(M=81, N=1280, K=1280)), ("lin 324x1280x1280", dict(M=324, N=1280, K=1280)),
         ("lin 324x1280x5120", dict(M=324, N=1280, K=5120)), ("lin 324x3840x1280", dict(M=324, N=3840, K=1280)),
         ("lin 1296x640x640", dict(M=1296, N=640, K=640)), ("lin 1296x640x2560", dict(M=1296, N=640, K=2560)),
         ("lin 257x1024x4096", dict(M=257, N=1024, K=4096)), ("lin 257x4096x1024", dict(M=257, N=4096, K=1024))]
if len(sys.argv) > 1 and sys.argv[1] == "mid":      # the 100-250-tile shapes of levels 0 / 1 (72x72 and 36x36 latents)
    probs = [("conv320@72x72", dict(N=320, conv=dict(T=1, H=72, W=72, C0=320, C1=0, kt=1, k=3))),
             ("conv640cat@72x72", dict(N=320, conv=dict(T=1, H=72, W=72, C0=320, C1=320, kt=1, k=3))),
             ("conv960cat@72x72", dict(N=320, conv=dict(T=1, H=72, W=72, C0=640, C1=320, kt=1, k=3))),
             ("conv1280cat@36x36", dict(N=640, conv=dict(T=1, H=36, W=36, C0=640, C1=640, kt=1, k=3))),
             ("conv1920cat@36x36", dict(N=640, conv=dict(T=1, H=36, W=36, C0=1280, C1=640, kt=1, k=3))),
             ("lin 5184x320x320", dict(M=5184, N=320, K=320)), ("lin 5184x960x320", dict(M=5184, N=960, K=320)),
             ("lin 5184x320x1280", dict(M=5184, N=320, K=1280)), ("lin 1296x1920x640", dict(M=1296, N=1920, K=640)),
             ("lin 1296x640x2560", dict(M=1296, N=640, K=2560)), ("lin 324x1280x5120", dict(M=324, N=1280, K=5120))]
for name, kw in probs:
    row = []
    for cfg in ((0, 1, 3, 12) if "ws" not in sys.argv else (0, 3, 19, 59, 63)):      # "ws": include the producer / consumer tiles (round 3)
        for sp in (1, 2, 4, 8, 12, 16, 24, 32):
            best = 1e9
            for r in range(2):
                try:
                    ms, tf, c, s = eng.bench_gemm(cfg=cfg, split=sp, iters=10, **kw)
                    best = min(best, ms * 1e3)
                except RuntimeError:
                    pass
            row.append((best, f"c{cfg}/s{sp}"))
    row.sort()
    ms, tf, c, s = eng.bench_gemm(iters=10, **kw)
    print(f"{name:20s} auto c{c}/s{s} {ms * 1e3:6.1f} us | " + "  ".join(f"{k}:{v:5.1f}" for v, k in row[:8]), flush=True)
