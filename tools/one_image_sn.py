#!/usr/bin/env python
"""One StableNormal call on one 576 x 576 image (BASELINE configs[3]) - the workload of the StableNormal counter-collection passes
(tools/pmc_traffic.sh sn).  usage: one_image_sn.py [--events OUT.json]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd.stablenormal import StableNormalPredictorHIP
H = W = 576
pred = StableNormalPredictorHIP.from_random(seed=7, workspace_bytes=24 << 30)
x = np.random.default_rng(0).uniform(0, 1, (1, H, W, 3)).astype(np.float32)
eng = pred.engine
if "--events" in sys.argv:
    pred.predict_batch(x)
    eng.profile_begin()
pred.predict_batch(x)
if "--events" in sys.argv:
    prof = eng.profile_end()
    gem = {k: v for k, v in prof.items() if k.startswith("gemm_")}
    calls = sum(v["calls"] for v in gem.values())
    json.dump({"denoise_steps": "stablenormal: 1 image 576x576, YOSO + 10 refinement steps", "gemm_launches": calls,
               "algorithmic_bytes_per_launch": sum(v["bytes"] for v in gem.values()) / calls,
               "algorithmic_tflop": sum(v["flops"] for v in gem.values()) / 1e12, "gemm_ms": sum(v["ms"] for v in gem.values())},
              open(sys.argv[sys.argv.index("--events") + 1], "w"))
