#!/usr/bin/env python
"""Time per K-step of the 192 x 128 producer / consumer tile (cfg 63) as a function of how many CUs are busy: the same N = 640 problem with M chosen so that
8 ... 500 tiles exist, K = 1280 and 2560 (W stays inside one L2; the difference isolates the K loop from launch / prologue / epilogue).  If a K-step gets faster when fewer CUs
run, the bound is a shared resource (L2 / fabric), if not it sits inside the CU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=24 << 30, persist_bytes=64 << 20)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 63
knobs = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # GEMM knob mask (2 = no XCD remap, 32 = row-major tile walk)
eng.tune_force(-100 - knobs, 0)
print(f"cfg {cfg} knobs {knobs}: M rows -> tiles (192 x 128 tiles, N = 640), us at K = 1280 / 2560, cycles per K-step per CU at 2.05 GHz")
for mt in (1, 2, 4, 8, 16, 32, 51, 100):          # M tiles; 5 N tiles each
    M = 192 * mt
    t = []
    for K in (1280, 2560):
        t.append(min(eng.bench_gemm(M=M, N=640, K=K, cfg=cfg, split=1, iters=20)[0] for _ in range(3)) * 1e3)
    tiles = mt * 5
    per_cu = -(-tiles // 256)
    step_us = (t[1] - t[0]) / (20 * per_cu)
    print(f"M {M:6d}  tiles {tiles:4d} (max {per_cu} per CU): {t[0]:7.1f} / {t[1]:7.1f} us  -> {step_us * 2050:7.0f} cycles per K-step", flush=True)
