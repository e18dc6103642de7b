#!/usr/bin/env python
"""Self-consistency of the GroupNorm statistics epilogue (knob 131072 = off) through the whole pipeline at several clip geometries, including ones whose
levels do not tile (the planner then declines per launch and GroupNorm runs its statistics pass): frames / depth of the two runs and ms per clip.
usage: ab_stats_epilogue.py [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
from unigeo_amd.synthetic import synthetic_clip
from unigeo_amd.model.depthcrafter import DepthCrafter
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
pipe = DepthCrafterPipelineHIP.from_random(seed=42, workspace_bytes=40 << 30)
eng = pipe.engine
for T, H, W in [(25, 384, 512), (10, 320, 448), (25, 448, 576), (7, 256, 320), (16, 576, 768)]:
    clip = synthetic_clip(T, H, W)
    nl, na = make_noise(T, H, W, 0)
    eng.set_inputs(DepthCrafter.prepare_input(None, clip), nl, na, np.stack(clip["intrinsics"], 0))
    out = {}
    for knob in (131072, 0):
        eng.tune_force(-100 - knob, -1)
        eng.run(steps, 8)
        t0 = time.perf_counter(); eng.run(steps, 8); dt = time.perf_counter() - t0
        fr, de, _ = eng.get_outputs()
        out[knob] = (fr.copy(), de.copy(), dt)
    eng.tune_force(-100 - 0, -1)
    df = np.abs(out[0][0] - out[131072][0]); dd = np.abs(out[0][1] - out[131072][1])
    print(f"{T:3d} x {H} x {W}, {steps} steps: frames max |diff| {df.max():.2e} mean {df.mean():.2e} | depth max {dd.max():.2e} mean {dd.mean():.2e} | "
          f"{out[131072][2] * 1e3:7.1f} ms with the statistics pass, {out[0][2] * 1e3:7.1f} ms with the epilogue statistics", flush=True)
