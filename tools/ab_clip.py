#!/usr/bin/env python
"""In-process A/B of engine switches on the full-size clip (25 x 384 x 512, 25 steps): box-to-box spread is +-4 %, so variants are
only ever compared inside one process.  usage: ab_clip.py epipre|walk|rowmajor|ff_fused|ff_ln|conv_split|fp8|vae32|flash|flashlazy|flashpp|group|snsmall|insitu|lvl3|ffxt|halo|halo_l0|lanes|lanes2|lanes4 [repeats]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
from unigeo_amd.synthetic import synthetic_clip
from unigeo_amd.model.depthcrafter import DepthCrafter
what = sys.argv[1] if len(sys.argv) > 1 else "ff_fused"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
T, H, W = 25, 384, 512
pipe = DepthCrafterPipelineHIP.from_random(seed=42, workspace_bytes=40 << 30)
eng = pipe.engine
clip = synthetic_clip(T, H, W)
nl, na = make_noise(T, H, W, 0)
eng.set_inputs(DepthCrafter.prepare_input(None, clip), nl, na, np.stack(clip["intrinsics"], 0))
if what.startswith("knob:"):      # generic: on = default, off = the knob mask set (a knob is an OFF / old-rule switch): ab_clip.py knob:16777216
    _mask = int(what.split(":")[1])
setter = {"knob:%d" % (_mask if what.startswith("knob:") else 0): (lambda on: eng.tune_force(-100 - (0 if on else _mask), 0)), "epipre": lambda on: eng.tune_force(-100 - (0 if on else 2097152), 0),     # round 5: epilogue operands prefetched during the K loop (192-row producer / consumer tiles)
          "walk": lambda on: eng.tune_force(-100 - (0 if on else 128), 0),           # round 5: XCD-owned tile runs (on) vs the round-strided walk
          "rowmajor": lambda on: eng.tune_force(-100 - (0 if on else 1048576), 0),  # round 5: row-major walk when a tile group fits the XCD's window anyway
          "conv_split": lambda on: eng.tune_force(-100 - (0 if on else 1024), 0), "group": lambda on: eng.tune_force(-100 - (0 if on else 32), 0),
          "flash": lambda on: eng.tune_flash(7 if on else 3), "flashlazy": lambda on: eng.tune_flash(23 if on else 7), "flashpp": lambda on: eng.tune_flash(87 if on else 23),
          "snsmall": lambda on: eng.tune_force(-100 - (0 if on else 8192), 0), "insitu": lambda on: eng.tune_force(-100 - (0 if on else 4096), 0),
          "lvl3": lambda on: eng.tune_force(-100 - (0 if on else 2048), 0), "ffxt": lambda on: eng.tune_ff(0 if on else 1),
          "halo": lambda on: eng.tune_force(-100 - (0 if on else 16384), 0), "halo_l0": lambda on: eng.tune_force(-100 - (0 if on else 32768), 0),
          "ff_fused": lambda on: eng.set_ff_fused(on, prenorm=on), "ff_ln": lambda on: eng.set_ff_fused(True, prenorm=on), "fp8": eng.set_fp8_linears, "vae32": eng.set_vae_encode_fp32,
          "lanes": lambda on: eng.set_concurrency(3 if on else 1), "lanes2": lambda on: eng.set_concurrency(2 if on else 1), "lanes4": lambda on: eng.set_concurrency(4 if on else 1)}[what]
eng.run(25, 8)
for rnd in range(reps):
    for on in (False, True):
        setter(on)
        eng.run(25, 8)
        t0 = time.perf_counter(); eng.run(25, 8); eng.run(25, 8); dt = (time.perf_counter() - t0) / 2
        print(f"round {rnd} {what}={int(on)}: {dt * 1e3:8.1f} ms/clip  {T / dt:6.2f} frames/s", flush=True)
