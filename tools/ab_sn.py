#!/usr/bin/env python
"""In-process A/B of a GEMM planner knob on one StableNormal image (576x576, batch 1; BASELINE configs[3]).
usage: ab_sn.py <knob bit, e.g. 512> [repeats]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd.stablenormal import StableNormalPredictorHIP
knob = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pred = StableNormalPredictorHIP.from_random(seed=7, workspace_bytes=24 << 30)
x = np.random.default_rng(0).uniform(0, 1, (1, 576, 576, 3)).astype(np.float32)
lib = pred.engine.lib
for _ in range(3): pred.predict_batch(x)
for r in range(reps):
    for off in (1, 0):
        pred.engine.tune_force(-100 - (knob if off else 0), 0)
        pred.predict_batch(x)
        t0 = time.perf_counter()
        for _ in range(5): pred.predict_batch(x)
        dt = (time.perf_counter() - t0) / 5
        print(f"round {r} knob {knob} {'set (rule off)' if off else 'clear (rule on)'}: {dt * 1e3:7.2f} ms/image  {1 / dt:6.2f} images/s", flush=True)
