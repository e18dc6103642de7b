#!/usr/bin/env python
"""In-process A/B of an engine switch on one StableNormal image (576x576, batch 1).  usage: ab_sn_gn.py [gn|lanes]
gn = one-launch GroupNorm on / off; lanes = guidance branch (DINO + DINO ControlNet) on a second stream beside YOSO vs in order."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd.stablenormal import StableNormalPredictorHIP
what = sys.argv[1] if len(sys.argv) > 1 else "gn"
pred = StableNormalPredictorHIP.from_random(seed=7, workspace_bytes=24 << 30)
x = np.random.default_rng(0).uniform(0, 1, (1, 576, 576, 3)).astype(np.float32)
for _ in range(3): pred.predict_batch(x)
for r in range(3):
    for on in (False, True):
        if what == "gn": pred.engine.set_gn_fused(on)
        else: pred.engine.set_concurrency(2 if on else 1)
        pred.predict_batch(x)
        t0 = time.perf_counter()
        for _ in range(5): pred.predict_batch(x)
        dt = (time.perf_counter() - t0) / 5
        print(f"round {r} {what}={int(on)}: {dt * 1e3:7.2f} ms/image  {1 / dt:6.2f} images/s", flush=True)
