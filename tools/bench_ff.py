#!/usr/bin/env python
"""A/B of the fused GEGLU feed-forward kernel (kernels/ff_fused.hip) against the two GEMM launches it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=8 << 30, persist_bytes=64 << 20)
for M, C in [(76800, 320), (19200, 320), (8192, 320), (76800, 256), (76800, 128)]:
    fl = 2.0 * M * (8 * C) * C + 2.0 * M * C * (4 * C)
    eng.bench_ff(M, C, True)                     # cold run of the shape: discard
    eng.tune_ff(1); a8 = min(eng.bench_ff(M, C, True) for _ in range(2))      # round-2 kernel (no cross-tile prefetch)
    eng.tune_ff(0); ax = min(eng.bench_ff(M, C, True) for _ in range(2))      # cross-tile prefetch (default)
    b = eng.bench_ff(M, C, False)
    print(f"M={M:6d} C={C:4d}: fused(cross-tile) {ax:8.1f} us {fl / ax / 1e6:7.0f} TF/s | fused(round 2) {a8:8.1f} us {fl / a8 / 1e6:7.0f} TF/s | "
          f"two launches {b:8.1f} us {fl / b / 1e6:7.0f} TF/s | x{b / ax:.2f}", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "ablate":
    # (needs the experiments build: make -C unigeo_amd/csrc experiments; UG_LIB_PATH=unigeo_amd/csrc/build/exp/libunigeo_exp.so)  Where does a packet's time go?  Timing-only variants of the round-2 kernel with one ingredient removed (results are wrong by design).
    M, C = 76800, 320
    names = {0: "full kernel", 1: "no GEGLU arithmetic", 2: "no weight-packet loads", 4: "no MFMAs", 8: "no fragment reads", 16: "no per-packet barrier",
             3: "no GEGLU, no loads", 10: "no loads, no fragment reads", 12: "no MFMAs, no fragment reads", 14: "loads off, MFMAs off, reads off", 18: "no loads, no barrier",
             30: "only GEGLU + prologue/epilogue left", 31: "only prologue/epilogue left"}
    eng.tune_ff(100); eng.bench_ff(M, C, True)
    for a in (0, 1, 2, 4, 8, 16, 3, 10, 12, 14, 18, 30, 31):
        eng.tune_ff(100 + a)
        us = min(eng.bench_ff(M, C, True) for _ in range(3))
        print(f"ablate {a:2d} ({names[a]:40s}): {us:7.1f} us", flush=True)
    eng.tune_ff(0)
