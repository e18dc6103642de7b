#!/usr/bin/env python
"""One full-size clip (25 x 384 x 512) with N denoise steps - the workload of the counter-collection passes (tools/pmc_traffic.sh).
usage: one_clip.py [steps] [--events OUT.json]   (--events: HIP-event profile of the same run -> algorithmic bytes / FLOPs per GEMM launch)"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
from unigeo_amd.synthetic import synthetic_clip
from unigeo_amd.model.depthcrafter import DepthCrafter
T, H, W = 25, 384, 512
steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
pipe = DepthCrafterPipelineHIP.from_random(seed=42, workspace_bytes=40 << 30)
clip = synthetic_clip(T, H, W)
nl, na = make_noise(T, H, W, 0)
pipe.engine.set_inputs(DepthCrafter.prepare_input(None, clip), nl, na, np.stack(clip["intrinsics"], 0))
if "--events" in sys.argv:
    pipe.engine.profile_begin()
pipe.engine.run(steps, 8)
if "--events" in sys.argv:
    prof = pipe.engine.profile_end()
    gem = {k: v for k, v in prof.items() if k.startswith("gemm_")}
    calls = sum(v["calls"] for v in gem.values())
    json.dump({"denoise_steps": steps, "gemm_launches": calls, "algorithmic_bytes_per_launch": sum(v["bytes"] for v in gem.values()) / calls,
               "algorithmic_tflop": sum(v["flops"] for v in gem.values()) / 1e12, "gemm_ms": sum(v["ms"] for v in gem.values())},
              open(sys.argv[sys.argv.index("--events") + 1], "w"))
