#!/usr/bin/env python
"""GroupNorm launch-scheme A/B on the shapes of the UNet / VAE (T, HW, C0, C1, pooled-over-frames).
mode 1 = stats / finalize / apply, 2 = one workgroup per (group, frame), 3 = one launch with the rows kept in registers (round 3; falls back to 1
where it is not eligible), 4 = slab in registers (one workgroup per group; falls back where the slab does not fit), 6 = pooled statistics from per-frame slabs in two launches (round 4), 0 = the launcher's pick.  Times are us per GroupNorm (back-to-back launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
eng = Engine(0, workspace_bytes=16 << 30, persist_bytes=64 << 20)
shapes = [(25, 3072, 320, 0, 0), (25, 3072, 320, 0, 1), (25, 3072, 320, 320, 0), (25, 3072, 640, 320, 0), (25, 768, 320, 0, 0), (25, 768, 640, 0, 0),
          (25, 768, 640, 0, 1), (25, 768, 640, 640, 0), (25, 768, 1280, 640, 0), (25, 768, 640, 320, 0), (25, 192, 640, 0, 0),
          (25, 192, 1280, 0, 0), (25, 192, 1280, 0, 1), (25, 192, 1280, 1280, 0), (25, 192, 1280, 640, 0), (25, 48, 1280, 0, 0),
          (25, 48, 1280, 0, 1), (25, 48, 1280, 1280, 0), (25, 192, 1280, 1280, 1), (25, 48, 1280, 1280, 1), (25, 768, 640, 640, 1), (1, 4096, 1280, 0, 0), (1, 1024, 1280, 0, 0), (1, 1024, 1280, 1280, 0), (1, 256, 1280, 0, 0), (1, 4096, 640, 0, 0), (1, 16384, 320, 0, 0), (8, 196608, 128, 0, 0), (8, 196608, 128, 0, 1), (8, 49152, 256, 0, 0),
          (8, 12288, 512, 0, 0), (8, 3072, 512, 0, 0), (8, 3072, 512, 0, 1)]
print("T      HW     C0    C1  pooled |   auto  3-launch   small  1-launch     slab  2-slabT")
for (T, HW, C0, C1, tp) in shapes:
    row = []
    for mode in (0, 1, 2, 3, 4, 6):
        best = 1e9
        for r in range(3):
            best = min(best, eng.bench_groupnorm(C0, C1, T, HW, tp, mode, iters=20))
        row.append(best)
    print(f"{T:3d} {HW:7d} {C0:5d} {C1:5d}  {tp:5d}  | " + "  ".join(f"{v:7.1f}" for v in row), flush=True)
