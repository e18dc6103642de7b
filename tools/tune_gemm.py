#!/usr/bin/env python
"""A/B the GEMM tile configurations / split-K factors on the problem shapes of one DepthCrafter clip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine

eng = Engine(0, workspace_bytes=24 << 30, persist_bytes=64 << 20)
if os.environ.get('UG_KNOBS'):
    eng.tune_force(-100 - int(os.environ['UG_KNOBS']), 0)
CFG = {0: "128x128x64s2", 1: "128x64x64s2", 2: "128x128x64s3", 3: "128x64x64s3", 4: "256x128x64s3", 5: "128x128x32s4",
       6: "256x256x32s3", 7: "256x128x32s4", 8: "256x128x64s2", 9: "128x128x32s3",
       10: "128x64x32s2", 11: "128x64x32s4", 12: "64x64x64s2", 13: "64x128x64s2", 14: "256x64x64s2", 15: "256x256x64s2(2x4)", 16: "256x256x64s2(4x2)", 17: "256x256x32s4", 18: "256x256x32s3", 19: "256x128x64s3(2x4)", 20: "256x128x32s2(4w)", 21: "128x256x32s2(4w)", 22: "256x128x32s3(4w)"}
dense = [(76800, 2560, 320), (76800, 320, 1280), (76800, 320, 320), (76800, 960, 320), (19200, 5120, 640),
         (19200, 640, 2560), (19200, 640, 640), (19200, 1920, 640), (4800, 10240, 1280), (4800, 1280, 5120), (4800, 1280, 1280),
         (4800, 3840, 1280), (1200, 10240, 1280), (1200, 1280, 5120), (1200, 1280, 1280), (6425, 5120, 1280), (6425, 1280, 5120), (8192, 8192, 8192)]
convs = [("vae128@384x512", 128, dict(T=8, H=384, W=512, C0=128, C1=0, kt=1, k=3)),
         ("vae256@192x256", 256, dict(T=8, H=192, W=256, C0=256, C1=0, kt=1, k=3)),
         ("vae512@96x128", 512, dict(T=8, H=96, W=128, C0=512, C1=0, kt=1, k=3)),
         ("vae512@48x64", 512, dict(T=8, H=48, W=64, C0=512, C1=0, kt=1, k=3)),
         ("unet320@48x64", 320, dict(T=25, H=48, W=64, C0=320, C1=0, kt=1, k=3)),
         ("unet640cat@48x64", 320, dict(T=25, H=48, W=64, C0=320, C1=320, kt=1, k=3)),
         ("unet1280cat@24x32", 640, dict(T=25, H=24, W=32, C0=640, C1=640, kt=1, k=3)),
         ("unet640@24x32", 640, dict(T=25, H=24, W=32, C0=640, C1=0, kt=1, k=3)),
         ("unet1280@12x16", 1280, dict(T=25, H=12, W=16, C0=1280, C1=0, kt=1, k=3)),
         ("unet1280@6x8", 1280, dict(T=25, H=6, W=8, C0=1280, C1=0, kt=1, k=3)),
         ("unet2560cat@6x8", 1280, dict(T=25, H=6, W=8, C0=1280, C1=1280, kt=1, k=3)),
         ("tconv320@48x64", 320, dict(T=25, H=48, W=64, C0=320, C1=0, kt=3, k=1)),
         ("tconv640@24x32", 640, dict(T=25, H=24, W=32, C0=640, C1=0, kt=3, k=1)),
         ("tconv1280@12x16", 1280, dict(T=25, H=12, W=16, C0=1280, C1=0, kt=3, k=1)),
         ("tconv512@96x128", 512, dict(T=8, H=96, W=128, C0=512, C1=0, kt=3, k=1)),
         ("tconv1280@6x8", 1280, dict(T=25, H=6, W=8, C0=1280, C1=0, kt=3, k=1)),
         ("tconv128@384x512", 128, dict(T=8, H=384, W=512, C0=128, C1=0, kt=3, k=1))]
cfgs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 3, 4, 8, 12, 14, 15, 19]
splits = [1, 2, 4, 8]
print(CFG)
print("dense")
for (M, N, K) in dense:
    row = []
    for c in cfgs:
        try:
            ms, tf, _, _ = eng.bench_gemm(M, N, K, cfg=c, split=1, iters=10)
            row.append(f"c{c}:{tf:5.0f}")
        except RuntimeError as e:
            row.append(f"c{c}:  err")
    print(f"{M:7d}x{N:5d}x{K:5d}  " + "  ".join(row))
    if M * N <= 1200 * 10240:
        r2 = []
        for sp in splits[1:]:
            for c in (0, 1):
                ms, tf, _, _ = eng.bench_gemm(M, N, K, cfg=c, split=sp, iters=10)
                r2.append(f"c{c}/s{sp}:{tf:5.0f}")
        print("          split-K:  " + "  ".join(r2))
print("conv")
for (name, N, cv) in convs:
    row = []
    for c in cfgs:
        try:
            ms, tf, _, _ = eng.bench_gemm(N=N, conv=cv, cfg=c, split=1, iters=10)
            row.append(f"c{c}:{tf:5.0f}")
        except RuntimeError as e:
            row.append(f"c{c}:  err")
    print(f"{name:18s}  " + "  ".join(row))
    if cv["T"] * cv["H"] * cv["W"] <= 25 * 12 * 16:
        r2 = []
        for sp in splits[1:]:
            for c in (0, 1):
                ms, tf, _, _ = eng.bench_gemm(N=N, conv=cv, cfg=c, split=sp, iters=10)
                r2.append(f"c{c}/s{sp}:{tf:5.0f}")
        print("          split-K:  " + "  ".join(r2))
ms, tf, c, s = eng.bench_gemm(N=1280, conv=convs[6][2], iters=10)
print("auto plan for unet1280@6x8:", CFG.get(c, c), "split", s, f"{tf:.0f} TF/s")
