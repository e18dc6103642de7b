#!/usr/bin/env python
"""Weight-stationary streaming GEMM (kernels/gemm_stream.hip; knob 65536 = off) against the planner's tiled kernel on the short-K projections:
us per launch (dense random operands, no residual) and the algorithmic-byte rate.  usage: ab_stream.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=24 << 30, persist_bytes=64 << 20)
print(f"{'shape':22s} {'cfg':>4s} {'tiled us':>9s} {'TB/s':>6s} | {'stream us':>9s} {'TB/s':>6s}  ratio")
for (M, N) in [(76800, 320), (76800, 640), (76800, 960), (19200, 320), (5184, 320), (5184, 960)]:
    by = 2.0 * (M * 320 + N * 320 + M * N)
    eng.tune_force(-100 - 65536, -1)
    a = min(eng.bench_gemm(M=M, N=N, K=320, iters=30) for _ in range(3))
    eng.tune_force(-100 - 0, -1)
    b = min(eng.bench_gemm(M=M, N=N, K=320, iters=30) for _ in range(3))
    print(f"{M}x{N}x320".ljust(22) + f" {a[2]:4d} {a[0]*1e3:9.1f} {by/(a[0]*1e-3)/1e12:6.2f} | {b[0]*1e3:9.1f} {by/(b[0]*1e-3)/1e12:6.2f}  {b[0]/a[0]:.3f}", flush=True)
