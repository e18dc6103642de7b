#!/usr/bin/env python
"""Round 6, upper bound for "lanes inside the UNet" (VERDICT r5 next 4): ONE clip's worth of frames (25) processed as two independent half clips (13 + 12
frames) on two engine contexts / streams / host threads at the same time, against the 25-frame clip on one context.  The halves run the same per-frame work
as the clip's frame groups would on two lanes (spatial layers are per frame; the temporal layers see 13 / 12 frames instead of 25 - same FLOPs per frame up to
the temporal attention's T x T term) with ZERO joins: what this does not gain, lanes with ~40 fork / join pairs per UNet forward cannot gain either.
usage: half_clips_in_flight.py [reps=3]  (UG_COSCHED=1: co-scheduled heuristics on the half contexts)"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
from unigeo_amd.synthetic import synthetic_clip
from unigeo_amd.model.depthcrafter import DepthCrafter
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H, W = 384, 512


def ctx(T, seed):
    pipe = DepthCrafterPipelineHIP.from_random(seed=42, workspace_bytes=24 << 30)
    clip = synthetic_clip(T, H, W, seed=1234 + seed)
    nl, na = make_noise(T, H, W, seed)
    pipe.engine.set_inputs(DepthCrafter.prepare_input(None, clip), nl, na, np.stack(clip["intrinsics"], 0))
    pipe.engine.run(25, 8)
    return pipe.engine


full, ha, hb = ctx(25, 0), ctx(13, 1), ctx(12, 2)
for e in (full, ha, hb):
    e.run(25, 8)


def timed(fn):
    t0 = time.perf_counter(); fn(); return time.perf_counter() - t0


def both():
    th = [threading.Thread(target=lambda e=e: [e.run(25, 8) for _ in range(reps)]) for e in (ha, hb)]
    [t.start() for t in th]; [t.join() for t in th]


for rnd in range(2):
    t_full = timed(lambda: [full.run(25, 8) for _ in range(reps)]) / reps
    t_a = timed(lambda: [ha.run(25, 8) for _ in range(reps)]) / reps
    t_b = timed(lambda: [hb.run(25, 8) for _ in range(reps)]) / reps
    if os.environ.get("UG_COSCHED"):
        ha.set_coscheduled(True); hb.set_coscheduled(True)
    t_ab = timed(both) / reps
    ha.set_coscheduled(False); hb.set_coscheduled(False)
    print(f"round {rnd}: 25-frame clip {t_full * 1e3:7.1f} ms | halves one after the other {t_a * 1e3:6.1f} + {t_b * 1e3:6.1f} = {(t_a + t_b) * 1e3:7.1f} ms | "
          f"halves in flight {t_ab * 1e3:7.1f} ms = {t_full / t_ab:.3f} x the clip", flush=True)
