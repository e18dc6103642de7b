#!/usr/bin/env python
"""Round 5: aggregate frames/s of ONE GPU with n independent clips in flight (n engine contexts, n host threads, n HIP streams) against one clip at a time.
The headline metric stays one clip at a time (BASELINE configs[1]: "single 25-frame 384x512 clip"); clips are independent samples (eval.py:33-56), so a
deployment that cares about throughput only can keep two in flight - what that buys is the chip time of ramp-up / tail / under-filled launches.
usage: two_clips_in_flight.py [n=2] [clips_per_context=3]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
from unigeo_amd.synthetic import synthetic_clip
from unigeo_amd.model.depthcrafter import DepthCrafter
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
T, H, W = 25, 384, 512
engs = []
for i in range(n):
    pipe = DepthCrafterPipelineHIP.from_random(seed=42, workspace_bytes=24 << 30)
    clip = synthetic_clip(T, H, W, seed=1234 + i)
    nl, na = make_noise(T, H, W, i)
    pipe.engine.set_inputs(DepthCrafter.prepare_input(None, clip), nl, na, np.stack(clip["intrinsics"], 0))
    if os.environ.get("UG_TUNE_KNOBS"):     # GEMM knob mask (kernels/gemm.hip) for every context
        pipe.engine.tune_force(-100 - int(os.environ["UG_TUNE_KNOBS"]), -1)
    if os.environ.get("UG_COSCHED"):
        pipe.engine.set_coscheduled(True)
    if os.environ.get("UG_COSCHED_KNOBS"):   # extra knob bits on top of the co-scheduled ones (A/B)
        pipe.engine.tune_force(-100 - (4194304 | int(os.environ["UG_COSCHED_KNOBS"])), -1)
    if os.environ.get("UG_TUNE_SPLIT"):
        pipe.engine.tune_force(-1, int(os.environ["UG_TUNE_SPLIT"]))
    pipe.engine.run(25, 8)
    engs.append(pipe.engine)
t0 = time.perf_counter()
for _ in range(reps):
    engs[0].run(25, 8)
one = reps * T / (time.perf_counter() - t0)
def work(e):
    for _ in range(reps):
        e.run(25, 8)
th = [threading.Thread(target=work, args=(e,)) for e in engs]
t0 = time.perf_counter()
[t.start() for t in th]; [t.join() for t in th]
agg = n * reps * T / (time.perf_counter() - t0)
print(f"one clip at a time: {one:.2f} frames/s;  {n} clips in flight: {agg:.2f} frames/s aggregate ({agg / one:.3f} x)")
