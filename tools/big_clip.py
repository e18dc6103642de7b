#!/usr/bin/env python
"""BASELINE configs[4] geometry in fp16: one 50-frame 576x768 clip, few denoise steps - checks that the engine scales past
the benchmark clip (workspace, 32-bit offsets, T = 50 temporal attention) and reports time per step."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
from unigeo_amd.synthetic import synthetic_clip
from unigeo_amd.model.depthcrafter import DepthCrafter
T, H, W = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (50, 576, 768)))
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
window, overlap = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (0, 0)   # latent sliding windows (T > 64 needs them)
pipe = DepthCrafterPipelineHIP.from_random(seed=42, workspace_bytes=96 << 30)
eng = pipe.engine
clip = synthetic_clip(T, H, W)
nl, na = make_noise(T, H, W, 0)
eng.set_inputs(DepthCrafter.prepare_input(None, clip), nl, na, np.stack(clip["intrinsics"], 0))
eng.run(1, 8, window=window, overlap=overlap)
t0 = time.time(); eng.run(steps, 8, window=window, overlap=overlap); t1 = time.time()
t2 = time.time(); eng.run(2 * steps, 8, window=window, overlap=overlap); t3 = time.time()
frames, depth, _ = eng.get_outputs()
per_step = ((t3 - t2) - (t1 - t0)) / steps
print(f"{T}x{H}x{W}: {steps} steps {t1 - t0:.3f} s, {2 * steps} steps {t3 - t2:.3f} s -> {per_step * 1e3:.1f} ms per denoise step, "
      f"fixed (CLIP + VAE) {(t1 - t0 - steps * per_step) * 1e3:.0f} ms; 25-step clip = {(t1 - t0 - steps * per_step + 25 * per_step):.2f} s "
      f"= {T / (t1 - t0 - steps * per_step + 25 * per_step):.1f} frames/s; workspace peak {eng.workspace_peak() / 2**30:.1f} GiB")
assert np.isfinite(depth).all() and depth.shape == (T, H, W)
print("depth range", float(depth.min()), float(depth.max()))
