#!/usr/bin/env python
"""A/B of the fused GEGLU feed-forward kernel (kernels/ff_fused.hip) against the two GEMM launches it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=8 << 30, persist_bytes=64 << 20)
for M, C in [(76800, 320), (19200, 320), (8192, 320), (76800, 256), (76800, 128)]:
    fl = 2.0 * M * (8 * C) * C + 2.0 * M * C * (4 * C)
    a, b = eng.bench_ff(M, C, True), eng.bench_ff(M, C, False)
    print(f"M={M:6d} C={C:4d}: fused {a:8.1f} us {fl / a / 1e6:7.0f} TF/s | two launches {b:8.1f} us {fl / b / 1e6:7.0f} TF/s | x{b / a:.2f}")
