#!/usr/bin/env python
"""Wall time of DepthCrafter.forward(data) as reference eval.py:39 calls it (torch imported first) on the full-size synthetic clip, beside the bare ug_dc_run with resident inputs."""
import os, sys, time
import torch  # noqa: F401  (first, as eval.py does)
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd.pipeline import DepthCrafterPipelineHIP
from unigeo_amd.synthetic import synthetic_clip
from unigeo_amd.model.depthcrafter import DepthCrafter
T, H, W = 25, 384, 512
pipe = DepthCrafterPipelineHIP.from_random(seed=42, workspace_bytes=40 << 30)
plug = DepthCrafter.__new__(DepthCrafter)
plug.pipeline, plug.num_inference_steps, plug.seed, plug._calls, plug.device = pipe, 25, 0, 0, "hip:0"
data = synthetic_clip(T, H, W, seed=1234)
plug.forward(data); plug.forward(data)
ts = []
for _ in range(4):
    t0 = time.perf_counter(); out = plug.forward(data); ts.append(time.perf_counter() - t0)
print("forward(data) ms:", " ".join(f"{t * 1e3:.1f}" for t in ts), f"best {T / min(ts):.2f} frames/s")
eng = pipe.engine
tr = []
for _ in range(3):
    t0 = time.perf_counter(); eng.run(25, 8, with_normals=True); tr.append(time.perf_counter() - t0)
print("ug_dc_run (resident inputs, with normals) ms:", " ".join(f"{t * 1e3:.1f}" for t in tr), f"best {T / min(tr):.2f} frames/s")
