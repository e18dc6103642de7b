#!/usr/bin/env python
"""A/B of the grouped tile walk (tile_coord, pick_group_m) on the clip's dense shapes: knob 32 = row-major walk everywhere."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=24 << 30, persist_bytes=64 << 20)
dense = [(76800, 2560, 320), (76800, 320, 1280), (19200, 5120, 640), (19200, 640, 2560), (19200, 1920, 640), (19200, 640, 640),
         (4800, 10240, 1280), (4800, 1280, 5120), (4800, 1280, 1280), (4800, 3840, 1280), (6425, 5120, 1280), (6425, 1280, 5120),
         (6425, 3840, 1280), (8192, 8192, 8192)]
for (M, N, K) in dense:
    row = []
    for knob in (32, 0, 32, 0):
        eng.tune_force(-100 - knob, 0)
        ms, tf, c, s = eng.bench_gemm(M, N, K, iters=20)
        row.append(f"{'row-major' if knob else 'grouped  '} {ms * 1e3:7.1f} us {tf:5.0f} TF/s")
    print(f"{M:6d}x{N:5d}x{K:5d} cfg {c:2d}: " + " | ".join(row), flush=True)
eng.tune_force(-100, 0)
