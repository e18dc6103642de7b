#!/usr/bin/env python
"""Per-K-step cycle stamps (s_memtime) of two co-resident waves of the GEMM kernel.  Needs the instrumented build:
   make -C unigeo_amd/csrc trace  ->  unigeo_amd/csrc/build/trace/libunigeo_trace.so
usage: gemm_trace.py <problem> <cfg>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unigeo_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "build", "trace", "libunigeo_trace.so")
os.environ["UG_GEMM_TRACE"] = "1"
P = {"vae512": dict(N=512, conv=dict(T=8, H=96, W=128, C0=512, C1=0, kt=1, k=3)),
     "unet320": dict(N=320, conv=dict(T=25, H=48, W=64, C0=320, C1=0, kt=1, k=3)),
     "ff1l1": dict(M=19200, N=5120, K=640), "ff2l1": dict(M=19200, N=640, K=2560), "ff2l2": dict(M=4800, N=1280, K=5120), "sq8k": dict(M=8192, N=8192, K=8192),
     "sq4k": dict(M=4096, N=4096, K=4096), "l1sq": dict(M=19200, N=640, K=640), "l1qkv": dict(M=19200, N=1920, K=640),
     "l1tc": dict(N=640, conv=dict(T=25, H=24, W=32, C0=640, C1=0, kt=3, k=1)), "unet1280": dict(N=1280, conv=dict(T=25, H=12, W=16, C0=1280, C1=0, kt=1, k=3))}
eng = L.Engine(0, workspace_bytes=24 << 30, persist_bytes=64 << 20)
name, cfg = sys.argv[1], int(sys.argv[2])
ms, tf, c, s = eng.bench_gemm(cfg=cfg, split=1, iters=3, **P[name])
sys.stdout.flush()
print(f"{name} cfg {c}: {ms*1e3:.1f} us  {tf:.0f} TF/s (instrumented)")
