#!/usr/bin/env python
"""Level-0 3x3 convolution (76800 rows, 320 columns): forced single-launch tile configs, us per launch and us per tile-round per 256 rows -
what a tile of each kind costs without the row split.  cfg 60 = im2col 256x160, 70 = halo 256x160 (8x32 px), 73 = halo 192x160 (12x16 px),
71 = halo 256x128 (3 column tiles, 1/6 wasted), 72 = halo 192x128."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=24 << 30, persist_bytes=64 << 20)
for name, cv in [("320->320", dict(T=25, H=48, W=64, C0=320, C1=0, kt=1, k=3)), ("640->320", dict(T=25, H=48, W=64, C0=320, C1=320, kt=1, k=3))]:
    for cfg, bm, bn in [(60, 256, 160), (70, 256, 160), (73, 192, 160), (71, 256, 128), (72, 192, 128), (63, 192, 128)]:
        try:
            r = min(eng.bench_gemm(N=320, conv=cv, cfg=cfg, split=1, iters=20) for _ in range(3))
        except Exception as e:
            print(name, cfg, "failed", str(e)[:80]); continue
        tiles = (76800 // bm) * -(-320 // bn)
        rounds = -(-tiles // 256)
        print(f"{name} cfg {cfg} ({bm}x{bn}): {r[0]*1e3:7.1f} us  {r[1]:6.0f} TF/s  {tiles} tiles = {rounds} rounds -> {r[0]*1e3/rounds:6.1f} us per round, {r[0]*1e3/rounds*256/bm*160/bn:6.1f} us per 256x160-equivalent round", flush=True)
