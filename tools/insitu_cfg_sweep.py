#!/usr/bin/env python
"""In-situ tile-config sweep: run the full-size clip with every GEMM forced to one tile config (split-K off) and print,
per GEMM shape, the HIP-event time under each config next to the planner's own choice.  Unlike tools/tune_gemm.py
(cache-cold, isolated) the operands here are wherever the producing kernel left them (L2 / Infinity Cache / HBM)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
from unigeo_amd.synthetic import synthetic_clip
from unigeo_amd.model.depthcrafter import DepthCrafter

cfgs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 14, 15, 19]
knobs = int(sys.argv[2]) if len(sys.argv) > 2 else 0
force_split = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # round 3: the same sweep with every plain-epilogue GEMM sliced along K (in-situ split-K study)
steps = 2
T, H, W = 25, 384, 512
pipe = DepthCrafterPipelineHIP.from_random(seed=42, workspace_bytes=40 << 30)
eng = pipe.engine
clip = synthetic_clip(T, H, W)
frames = DepthCrafter.prepare_input(None, clip)
nl, na = make_noise(T, H, W, 0)
eng.set_inputs(frames, nl, na, np.stack(clip["intrinsics"], 0))
eng.run(1, 8)
res = {}
for c in [-1] + cfgs:
    eng.tune_force(-100 - knobs, 0)
    eng.tune_force(c, force_split if c >= 0 else -1)
    eng.run(1, 8)
    eng.profile_begin(shapes=True)
    eng.run(steps, 8)
    prof = eng.profile_end()
    for k, v in prof.items():
        if k.startswith("gemm_"):
            res.setdefault(k, {})[c] = (v["ms"] * 1000 / v["calls"], v["calls"], v["flops"] / v["calls"])
eng.tune_force(-1, -1)
rows = sorted(res.items(), key=lambda kv: -kv[1][-1][0] * kv[1][-1][1])
print("shape".ljust(44) + "calls  " + "  ".join(f"{'plan' if c < 0 else 'c%d' % c:>8s}" for c in [-1] + cfgs) + "   best")
for k, d in rows[:60]:
    best = min((v[0], c) for c, v in d.items() if c >= 0)
    print(k.ljust(44) + f"{d[-1][1]:5d}  " + "  ".join(f"{d[c][0]:8.1f}" if c in d else "       -" for c in [-1] + cfgs) +
          f"   c{best[1]} ({d[-1][0] / best[0]:.2f}x)")
