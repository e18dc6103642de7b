#!/usr/bin/env python
"""One full-size clip with a short denoise loop (3 steps) - the workload for counter-collection passes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
from unigeo_amd.synthetic import synthetic_clip
from unigeo_amd.model.depthcrafter import DepthCrafter
T, H, W = 25, 384, 512
pipe = DepthCrafterPipelineHIP.from_random(seed=42, workspace_bytes=40 << 30)
clip = synthetic_clip(T, H, W)
nl, na = make_noise(T, H, W, 0)
pipe.engine.set_inputs(DepthCrafter.prepare_input(None, clip), nl, na, np.stack(clip["intrinsics"], 0))
pipe.engine.run(int(sys.argv[1]) if len(sys.argv) > 1 else 3, 8)
