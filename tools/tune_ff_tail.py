#!/usr/bin/env python
"""Tile-config x split-K sweep for the two GEMMs that carry the feed-forward TAIL rows of level 0 (the 88 row tiles beyond two whole rounds of the fused
kernel: M = 11264): 11264 x 2560 x 320 with the GEGLU epilogue (UG_BENCH_GEGLU=1) and 11264 x 320 x 1280 with a residual (UG_BENCH_R1=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=8 << 30, persist_bytes=64 << 20)
which = sys.argv[1] if len(sys.argv) > 1 else "down"
if which == "down":
    os.environ["UG_BENCH_R1"] = "1"
    kw, cfgs, splits = dict(M=11264, N=320, K=1280), (0, 1, 3, 4, 14, 19, 59, 63, 60), (1, 2, 3, 4)
else:
    os.environ["UG_BENCH_GEGLU"] = "1"
    kw, cfgs, splits = dict(M=11264, N=2560, K=320), (0, 4, 8, 15, 35, 54, 62, 64), (1,)
row = []
for cfg in cfgs:
    for sp in splits:
        best = 1e9
        for r in range(3):
            try:
                ms, tf, c, s = eng.bench_gemm(cfg=cfg, split=sp, iters=20, **kw)
                best = min(best, ms)
            except RuntimeError:
                pass
        if best < 1e9:
            row.append((best * 1000, f"c{cfg}/s{sp}"))
row.sort()
ms, tf, c, s = eng.bench_gemm(iters=20, **kw)
print(f"{which} {kw}: auto c{c}/s{s} {ms * 1000:6.1f} us | " + "  ".join(f"{k}:{v:5.1f}" for v, k in row[:12]), flush=True)
