#!/usr/bin/env python
"""ms per full-size clip (25 x 384 x 512, 25 steps) of the current build under whatever UG_* environment knobs are set - for A/B runs of
load-time knobs (UG_GN_WANT ...) as consecutive processes on ONE box.  usage: time_clip.py [clips]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
from unigeo_amd.synthetic import synthetic_clip
from unigeo_amd.model.depthcrafter import DepthCrafter
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
T, H, W = 25, 384, 512
pipe = DepthCrafterPipelineHIP.from_random(seed=42, workspace_bytes=40 << 30)
eng = pipe.engine
if os.environ.get("UG_TUNE_KNOBS"):                       # GEMM knob mask for this run (kernels/gemm.hip), e.g. 16384 = halo conv off
    eng.tune_force(-100 - int(os.environ["UG_TUNE_KNOBS"]), -1)
clip = synthetic_clip(T, H, W)
nl, na = make_noise(T, H, W, 0)
eng.set_inputs(DepthCrafter.prepare_input(None, clip), nl, na, np.stack(clip["intrinsics"], 0))
eng.run(25, 8); eng.run(25, 8)
t0 = time.perf_counter()
for _ in range(n):
    eng.run(25, 8)
dt = (time.perf_counter() - t0) / n
print(f"{dt * 1e3:8.1f} ms/clip  {T / dt:6.2f} frames/s   env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("UG_")), flush=True)
