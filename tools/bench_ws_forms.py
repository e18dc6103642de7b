#!/usr/bin/env python
"""Isolated A/B of the producer / consumer GEMM forms: eight consumer waves (cfg 63 / 64 on the 192 x 128 tile, 59 / 54 on 256 x 128) against four consumer waves with
twice the wave tile (cfg 65 / 66).  usage: bench_ws_forms.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=24 << 30, persist_bytes=64 << 20)
shapes = [("19200x640x2560", dict(M=19200, N=640, K=2560), 63), ("4800x1280x5120", dict(M=4800, N=1280, K=5120), 63), ("19200x5120x640", dict(M=19200, N=5120, K=640), 64),
          ("19200x1920x640", dict(M=19200, N=1920, K=640), 63), ("19200x640x640", dict(M=19200, N=640, K=640), 63), ("4800x1280x1280", dict(M=4800, N=1280, K=1280), 63), ("4800x3840x1280", dict(M=4800, N=3840, K=1280), 63), ("8192^3", dict(M=8192, N=8192, K=8192), 63),
          ("conv640@24x32", dict(N=640, conv=dict(T=25, H=24, W=32, C0=640, C1=0, kt=1, k=3)), 63), ("conv1280@12x16", dict(N=1280, conv=dict(T=25, H=12, W=16, C0=1280, C1=0, kt=1, k=3)), 63)]
for name, kw, cfg in shapes:
    row = []
    for c in (63, 64, 65, 59, 54, 66, 35):
        best = min(eng.bench_gemm(cfg=c, split=1, iters=20, **kw)[0] for _ in range(3))
        row.append(best * 1e3)
    print(f"{name:18s}: c63 {row[0]:8.1f} | c64 {row[1]:8.1f} | c65 (4 consumers 192x128) {row[2]:8.1f} || c59 {row[3]:8.1f} | c54 {row[4]:8.1f} | c66 (4 consumers 256x128) {row[5]:8.1f} || c35 (256x256) {row[6]:8.1f}", flush=True)
