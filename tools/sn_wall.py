#!/usr/bin/env python
"""Wall-clock ms per StableNormal image at batch 1 (576 x 576) - run it alone and under `rocprofv3 --kernel-trace --stats` to compare the wall time with the
sum of the kernel durations (round 3: 90.1 ms wall against 88.6 ms of kernels over 7370 launches per image - the GPU, not the host, is the limit at batch 1)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd.stablenormal import StableNormalPredictorHIP
pred = StableNormalPredictorHIP.from_random(seed=7, workspace_bytes=24 << 30)
x = np.random.default_rng(0).uniform(0, 1, (1, 576, 576, 3)).astype(np.float32)
for _ in range(3): pred.predict_batch(x)
t0 = time.perf_counter()
for _ in range(5): pred.predict_batch(x)
print("wall ms/image", (time.perf_counter() - t0) / 5 * 1e3)
