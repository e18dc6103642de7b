#!/usr/bin/env python
"""Per-shape HIP-event profile of one StableNormal call (B images of 576x576; BASELINE configs[3])."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd.stablenormal import StableNormalPredictorHIP
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H = W = 576
pred = StableNormalPredictorHIP.from_random(seed=7, workspace_bytes=24 << 30)
x = np.random.default_rng(0).uniform(0, 1, (B, H, W, 3)).astype(np.float32)
pred.predict_batch(x)
eng = pred.engine
eng.profile_begin(shapes=True)
pred.predict_batch(x)
prof = eng.profile_end()
tot = sum(v["ms"] for v in prof.values())
print(f"total {tot:.1f} ms for {B} image(s)")
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:45]:
    tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] else 0
    print(f"{k:48s} {v['ms']:9.2f} ms {v['calls']:5d} calls {v['ms']*1000/v['calls']:9.1f} us/call {tf:8.1f} TF/s")
