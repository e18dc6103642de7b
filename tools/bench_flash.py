#!/usr/bin/env python
"""A/B of flash-attention variants on the clip's shapes (device-resident random data).
variant bits: 1 = one softmax step per 64 keys, 2 = XCD-grouped workgroup order, 4 = 2-slot ring + 4 workgroups per CU, 16 = lazy rescale + dot2 row sums (23 = default),
64 = 8-wave ping-pong kernel (87; + 128: priority flipped per phase, + 256: no priority).  `ablate` (experiments build: make -C unigeo_amd/csrc experiments; UG_LIB_PATH=unigeo_amd/csrc/build/exp/libunigeo_exp.so): timing-only ablations of variant 7; `one`: the level-0 shape only (profiler runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=8 << 30, persist_bytes=64 << 20)
if len(sys.argv) > 1 and sys.argv[1] == "one":          # profiler runs (tools/pmc_flash.sh): the level-0 shape only
    for _ in range(3): eng.bench_flash(25, 5, 3072, 23, iters=5)
    sys.exit(0)
for B, H, S in [(25, 5, 3072), (25, 10, 768), (25, 20, 192), (1, 5, 5184), (8, 5, 5184)]:
    fl = 4.0 * B * H * S * S * 64
    eng.bench_flash(B, H, S, 0)                                   # the first measurement of a shape runs cold (clock ramp): discard
    r = [min(eng.bench_flash(B, H, S, v) for _ in range(3)) for v in (7, 23, 87)]
    r = [min(a, eng.bench_flash(B, H, S, v)) for v, a in zip((7, 23, 87), r)]
    print(f"B={B:3d} H={H:3d} S={S:5d}: " + " | ".join(f"v{v}: {u:8.1f} us {fl / u / 1e6:6.0f} TF/s" for v, u in zip((7, 23, 87), r)))
if len(sys.argv) > 1 and sys.argv[1] == "ablate":
    # timing-only ablations of the default kernel (wrong results): what a tile's time is made of
    names = {7: "full", 1001: "no exp", 1002: "no QK MFMA", 1004: "no PV MFMA", 1006: "no MFMA", 1008: "no softmax arithmetic", 1014: "no MFMA, no softmax", 1007: "no exp, no MFMA"}
    B, H, S = 25, 5, 3072
    for v, nm in names.items():
        u = min(eng.bench_flash(B, H, S, v) for _ in range(4))
        print(f"  {nm:24s} {u:8.1f} us")
