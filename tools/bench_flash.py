#!/usr/bin/env python
"""A/B of flash-attention variants on the clip's shapes (device-resident random data).
variant bits: 1 = one softmax step per 64 keys, 2 = XCD-grouped workgroup order, 4 = 2-slot ring + 4 workgroups per CU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=8 << 30, persist_bytes=64 << 20)
for B, H, S in [(25, 5, 3072), (25, 10, 768), (25, 20, 192), (1, 5, 5184), (8, 5, 5184)]:
    fl = 4.0 * B * H * S * S * 64
    eng.bench_flash(B, H, S, 0)                                   # the first measurement of a shape runs cold (clock ramp): discard
    r = [min(eng.bench_flash(B, H, S, v) for _ in range(3)) for v in (3, 6, 7)]
    r = [min(a, eng.bench_flash(B, H, S, v)) for v, a in zip((3, 6, 7), r)]
    print(f"B={B:3d} H={H:3d} S={S:5d}: " + " | ".join(f"v{v}: {u:8.1f} us {fl / u / 1e6:6.0f} TF/s" for v, u in zip((3, 6, 7), r)))
