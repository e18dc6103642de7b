#!/usr/bin/env python
"""Calibration only (not on the product path): what the vendor GEMM (torch.matmul -> hipBLASLt) reaches on this box for
the dense problem shapes of one clip, cache-cold (rotating operands), to put the hand-written kernel's TFLOP/s in context."""
import torch
shapes = [(76800, 2560, 320), (76800, 320, 1280), (76800, 960, 320), (76800, 320, 320), (19200, 5120, 640), (19200, 640, 2560),
          (19200, 1920, 640), (4800, 10240, 1280), (4800, 1280, 5120), (4800, 3840, 1280), (6425, 5120, 1280), (6425, 1280, 5120),
          (8192, 8192, 8192), (4096, 4096, 4096)]
dev = "cuda"
for (M, N, K) in shapes:
    per = (M * K + M * N) * 2
    nbuf = max(2, min(16, (600 << 20) // per + 1))
    As = [torch.randn(M, K, device=dev, dtype=torch.float16) for _ in range(nbuf)]
    Os = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(nbuf)]
    W = torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5
    for i in range(3):
        torch.matmul(As[i % nbuf], W.t(), out=Os[i % nbuf])
    best = 0
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(10):
            torch.matmul(As[i % nbuf], W.t(), out=Os[i % nbuf])
        e1.record(); torch.cuda.synchronize()
        best = max(best, 2.0 * M * N * K / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12)
    print(f"{M}x{N}x{K}: hipBLASLt {best:6.0f} TF/s", flush=True)
    del As, Os, W
