#!/usr/bin/env python
"""Per-shape HIP-event profile of one full-size clip: which GEMM / conv / norm shapes the time goes to."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
from unigeo_amd.synthetic import synthetic_clip
from unigeo_amd.model.depthcrafter import DepthCrafter

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
T, H, W = 25, 384, 512
pipe = DepthCrafterPipelineHIP.from_random(seed=42, workspace_bytes=40 << 30)
eng = pipe.engine
clip = synthetic_clip(T, H, W)
frames = DepthCrafter.prepare_input(None, clip)
nl, na = make_noise(T, H, W, 0)
eng.set_inputs(frames, nl, na, np.stack(clip["intrinsics"], 0))
if os.environ.get("UG_TUNE_KNOBS"):     # GEMM knob mask for this run (kernels/gemm.hip), e.g. 16384 = halo conv off
    eng.tune_force(-100 - int(os.environ["UG_TUNE_KNOBS"]), -1)
if os.environ.get("UG_FP8"):
    eng.set_fp8_linears(True)           # BASELINE configs[4] option: MX-fp8 linear layers
if os.environ.get("UG_NO_FF_FUSED"):
    eng.set_ff_fused(False)
eng.run(1, 8)
eng.profile_begin(shapes=True)
eng.run(steps, 8)
prof = eng.profile_end()
rows = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for v in prof.values())
print(f"total {tot:.1f} ms over {steps} denoise steps")
for k, v in rows[:70]:
    tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] else 0
    gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["bytes"] else 0
    print(f"{k:48s} {v['ms']:9.2f} ms {v['calls']:5d} calls {v['ms']*1000/v['calls']:9.1f} us/call {tf:8.1f} TF/s {gb:8.0f} GB/s")
print("note: gemm_conv_up2x2 rows are credited the 9-tap algorithmic FLOPs of upsample+conv while executing 4 taps - not MFMA rates")
json.dump(prof, open("gpurun_out/shapes.json", "w"))
