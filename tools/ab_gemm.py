#!/usr/bin/env python
"""A/B GEMM variants on the problem shapes of one clip, interleaved rounds in one process (best of 3 x 10 launches,
cache-cold rotating operands).  usage: ab_gemm.py [variant,variant,...]; a variant is <cfg>[k<knobs>], cfg -1 = auto plan;
knobs: 1 one tile per workgroup, 2 no XCD remap, 4 flat (64-bit) addressing instead of buffer addressing.
UG_BENCH_GEGLU=1 switches the dense problems to the GEGLU epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine

eng = Engine(0, workspace_bytes=24 << 30, persist_bytes=64 << 20)
dense = [(76800, 2560, 320), (76800, 320, 1280), (76800, 960, 320), (76800, 320, 320), (19200, 5120, 640), (19200, 640, 2560),
         (19200, 1920, 640), (19200, 640, 640), (4800, 10240, 1280), (4800, 1280, 5120), (4800, 3840, 1280), (4800, 1280, 1280),
         (1200, 10240, 1280), (1200, 1280, 5120), (6425, 5120, 1280), (6425, 1280, 5120), (8192, 8192, 8192)]
convs = [("vae128@384x512", 128, dict(T=8, H=384, W=512, C0=128, C1=0, kt=1, k=3)),
         ("vae256@192x256", 256, dict(T=8, H=192, W=256, C0=256, C1=0, kt=1, k=3)),
         ("vae512@96x128", 512, dict(T=8, H=96, W=128, C0=512, C1=0, kt=1, k=3)),
         ("vae512@48x64", 512, dict(T=8, H=48, W=64, C0=512, C1=0, kt=1, k=3)),
         ("unet320@48x64", 320, dict(T=25, H=48, W=64, C0=320, C1=0, kt=1, k=3)),
         ("unet640cat@48x64", 320, dict(T=25, H=48, W=64, C0=320, C1=320, kt=1, k=3)),
         ("unet640@24x32", 640, dict(T=25, H=24, W=32, C0=640, C1=0, kt=1, k=3)),
         ("unet1280@12x16", 1280, dict(T=25, H=12, W=16, C0=1280, C1=0, kt=1, k=3)),
         ("unet1280@6x8", 1280, dict(T=25, H=6, W=8, C0=1280, C1=0, kt=1, k=3)),
         ("tconv320@48x64", 320, dict(T=25, H=48, W=64, C0=320, C1=0, kt=3, k=1)),
         ("tconv640@24x32", 640, dict(T=25, H=24, W=32, C0=640, C1=0, kt=3, k=1)),
         ("tconv1280@12x16", 1280, dict(T=25, H=12, W=16, C0=1280, C1=0, kt=3, k=1)),
         ("tconv512@96x128", 512, dict(T=8, H=96, W=128, C0=512, C1=0, kt=3, k=1)),
         ("tconv128@384x512", 128, dict(T=8, H=384, W=512, C0=128, C1=0, kt=3, k=1))]
VARS = (sys.argv[1] if len(sys.argv) > 1 else "-1,-1k4").split(",")


def parse(v):
    cfg, _, kn = v.partition("k")
    return int(cfg), int(kn or 0)


def ab(label, **kw):
    best = {}
    for rnd in range(3):
        for v in VARS:
            cfg, kn = parse(v)
            eng.tune_force(-100 - kn, 0)
            try:
                ms, tf, c, s = eng.bench_gemm(cfg=cfg, split=(0 if cfg < 0 else 1), iters=10, **kw)
            except RuntimeError:
                continue
            key = f"{v}(c{c}/s{s})" if cfg < 0 else v
            best[key] = max(best.get(key, 0), tf)
    eng.tune_force(-100, 0)
    print(f"{label:22s} " + "  ".join(f"{k}:{v:6.0f}" for k, v in best.items()), flush=True)


print("GEGLU epilogue" if os.environ.get("UG_BENCH_GEGLU") else "plain bias epilogue")
for (M, N, K) in dense:
    ab(f"{M}x{N}x{K}", M=M, N=N, K=K)
for (name, N, cv) in convs:
    ab(name, N=N, conv=cv)
