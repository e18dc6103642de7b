import os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from unigeo_amd.stablenormal import StableNormalPredictorHIP
pred = StableNormalPredictorHIP.from_random(seed=7, workspace_bytes=24 << 30)
x = np.random.default_rng(0).uniform(0, 1, (1, 576, 576, 3)).astype(np.float32)
for _ in range(3): pred.predict_batch(x)
for r in range(3):
    for on in (False, True):
        pred.engine.set_gn_fused(on)
        pred.predict_batch(x)
        t0 = time.perf_counter()
        for _ in range(5): pred.predict_batch(x)
        dt = (time.perf_counter() - t0) / 5
        print(f"round {r} gn_fused={int(on)}: {dt * 1e3:7.2f} ms/image  {1 / dt:6.2f} images/s", flush=True)
