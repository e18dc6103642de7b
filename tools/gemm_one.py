#!/usr/bin/env python
"""Run one GEMM / implicit-GEMM problem a few times (for rocprofv3 counter passes).  usage: gemm_one.py <name> <cfg> [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd._lib import Engine
P = {"vae512": dict(N=512, conv=dict(T=8, H=96, W=128, C0=512, C1=0, kt=1, k=3)),
     "unet320": dict(N=320, conv=dict(T=25, H=48, W=64, C0=320, C1=0, kt=1, k=3)),
     "ff1l1": dict(M=19200, N=5120, K=640), "ff2l1": dict(M=19200, N=640, K=2560), "ff2l2": dict(M=4800, N=1280, K=5120), "sq8k": dict(M=8192, N=8192, K=8192),
     "qkvl1": dict(M=19200, N=1920, K=640), "ff1l2": dict(M=4800, N=10240, K=1280)}
eng = Engine(0, workspace_bytes=24 << 30, persist_bytes=64 << 20)
name, cfg = sys.argv[1], int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
if os.environ.get("UG_KNOBS"):      # GEMM knob mask (kernels/gemm.hip), e.g. 128 = the round-strided XCD walk of rounds 1 - 4
    eng.tune_force(-100 - int(os.environ["UG_KNOBS"]), 0)
ms, tf, c, s = eng.bench_gemm(cfg=cfg, split=1, iters=iters, **P[name])
print(f"{name} cfg {c}: {ms*1e3:.1f} us  {tf:.0f} TF/s")
