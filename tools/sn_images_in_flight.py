#!/usr/bin/env python
"""Round 5: StableNormal (BASELINE configs[3]) with n independent images in flight on ONE GPU - n predictor contexts, n host threads, one 576 x 576 image per call
each (the reference calls the predictor once per frame, model/stablenormal.py:39; frames are independent).  usage: sn_images_in_flight.py [n=3] [images per context=6]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd.stablenormal import StableNormalPredictorHIP
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
H = W = 576
yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
img = np.clip(np.stack([127.5 + 100 * np.sin(xx / 41.0 + c) * np.cos(yy / 29.0) for c in range(3)], -1), 0, 255).astype(np.uint8).astype(np.float32)[None] / 255.0
if os.environ.get("UG_SN_LATE"):     # A/B: the first context created (with a 24 GB workspace) and used before the others exist, as bench.py does
    preds = [StableNormalPredictorHIP.from_random(seed=7, workspace_bytes=int(os.environ["UG_SN_LATE"]) << 30)]
    for _ in range(8):
        preds[0].predict_batch(img)
    preds += [StableNormalPredictorHIP.from_random(seed=7, workspace_bytes=8 << 30) for _ in range(n - 1)]
else:
    preds = [StableNormalPredictorHIP.from_random(seed=7, workspace_bytes=8 << 30) for _ in range(n)]
for p in preds:
    if os.environ.get("UG_COSCHED"):
        p.engine.set_coscheduled(True)
    for _ in range(2):
        p.predict_batch(img)
t0 = time.perf_counter()
for _ in range(reps):
    preds[0].predict_batch(img)
one = reps / (time.perf_counter() - t0)
def work(p):
    for _ in range(reps):
        p.predict_batch(img)
th = [threading.Thread(target=work, args=(p,)) for p in preds]
t0 = time.perf_counter()
[t.start() for t in th]; [t.join() for t in th]
agg = n * reps / (time.perf_counter() - t0)
print(f"StableNormal 576x576: one image at a time {one:.2f} images/s;  {n} images in flight: {agg:.2f} images/s aggregate ({agg / one:.2f} x)")
