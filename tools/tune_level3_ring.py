#!/usr/bin/env python
"""Round 6: does a deeper ring help where the operands are first-touch HBM bytes?  The 192 x 128 producer / consumer tile (configs 63 / 64) on the level-3 and level-2
problems, split-K 1 .. 6, for whatever library UG_LIB_PATH selects (the -DUG_WS_NST4 build has four 40-KiB slots on that tile instead of three), beside the planner's pick."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=8 << 30, persist_bytes=64 << 20)
probs = [("conv1280@6x8", dict(N=1280, conv=dict(T=25, H=6, W=8, C0=1280, C1=0, kt=1, k=3))),
         ("tconv1280@6x8", dict(N=1280, conv=dict(T=25, H=6, W=8, C0=1280, C1=0, kt=3, k=1))),
         ("lin 1200x1280x5120", dict(M=1200, N=1280, K=5120)), ("lin 1200x10240x1280", dict(M=1200, N=10240, K=1280)),
         ("lin 4800x1280x5120", dict(M=4800, N=1280, K=5120)), ("lin 4800x3840x1280", dict(M=4800, N=3840, K=1280)),
         ("lin 19200x640x2560", dict(M=19200, N=640, K=2560)), ("lin 19200x1920x640", dict(M=19200, N=1920, K=640)),
         ("tconv1280@12x16", dict(N=1280, conv=dict(T=25, H=12, W=16, C0=1280, C1=0, kt=3, k=1)))]
for name, kw in probs:
    row = []
    for cfg in (63, 64):
        for sp in (1, 2, 3, 4, 6):
            best = 1e9
            for r in range(3):
                try:
                    ms, tf, c, s = eng.bench_gemm(cfg=cfg, split=sp, iters=10, **kw)
                    best = min(best, ms)
                except RuntimeError:
                    pass
            if best < 1e9:
                row.append((best * 1000, f"c{cfg}/s{sp}"))
    row.sort()
    ms, tf, c, s = eng.bench_gemm(iters=10, **kw)
    print(f"{name:20s} auto c{c}/s{s} {ms * 1000:6.1f} us | " + "  ".join(f"{k}:{v:5.1f}" for v, k in row[:6]), flush=True)
