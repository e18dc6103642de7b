#!/usr/bin/env python
"""Fused LayerNorm -> Q|K|V projection kernel against the LayerNorm launch + GEMM it replaces (device time per call, level-0 shape of the clip)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=8 << 30, persist_bytes=64 << 20)
rng = np.random.default_rng(0)
for M, C, N in [(76800, 320, 960), (65536, 320, 960), (76800, 320, 320), (19200, 320, 960)]:
    X = rng.standard_normal((M, C)).astype(np.float32); g = np.ones(C, np.float32); b = np.zeros(C, np.float32)
    W = (rng.standard_normal((N, C)) / np.sqrt(C)).astype(np.float32)
    _, a = eng.op_ln_linear(X, g, b, W, fused=True, iters=20)
    _, t = eng.op_ln_linear(X, g, b, W, fused=False, iters=20)
    fl = 2.0 * M * N * C
    print(f"M={M:6d} C={C} N={N:4d}: fused {a:7.1f} us {fl / a / 1e6:6.0f} TF/s | LayerNorm + GEMM {t:7.1f} us | x{t / a:.2f}", flush=True)
