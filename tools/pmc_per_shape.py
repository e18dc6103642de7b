#!/usr/bin/env python
"""Per-shape HBM traffic of the GEMM kernels: joins rocprofv3's per-dispatch FETCH_SIZE / WRITE_SIZE rows (dispatch order) with the
engine's own ordered list of GEMM launches (names carry the shape; algorithmic bytes per launch).
usage (GPU box): python tools/pmc_per_shape.py [denoise_steps]    -> gpurun_out/pmc_per_shape.txt"""
import collections, csv, glob, json, os, subprocess, sys
steps = sys.argv[1] if len(sys.argv) > 1 else "2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs("gpurun_out", exist_ok=True)
RX = "gemm_(kernel|ldr_kernel|ws_kernel)|ff_fused_kernel"
if len(sys.argv) > 2 and sys.argv[2] == "--child":      # the profiled workload: one clip, shape-level event profile with launch order
    sys.path.insert(0, root)
    import numpy as np
    from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
    from unigeo_amd.synthetic import synthetic_clip
    from unigeo_amd.model.depthcrafter import DepthCrafter
    T, H, W = 25, 384, 512
    pipe = DepthCrafterPipelineHIP.from_random(seed=42, workspace_bytes=40 << 30)
    clip = synthetic_clip(T, H, W); nl, na = make_noise(T, H, W, 0)
    eng = pipe.engine
    eng.set_inputs(DepthCrafter.prepare_input(None, clip), nl, na, np.stack(clip["intrinsics"], 0))
    eng.profile_begin(shapes=True)
    eng.run(int(steps), 8)
    eng.profile_end()
    if os.environ.get("UG_DUMP_ORDER"):
        json.dump(eng.last_profile_order, open(os.environ["UG_DUMP_ORDER"], "w"))
    sys.exit(0)
env = dict(os.environ, TMPDIR="/tmp")
subprocess.run([sys.executable, __file__, steps, "--child"], env=dict(env, UG_DUMP_ORDER="gpurun_out/pmc_order.json"), check=True)
order = json.load(open("gpurun_out/pmc_order.json"))
vals = {}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    d = f"gpurun_out/pmcps_{cname}"
    subprocess.run(["rm", "-rf", d])
    subprocess.run(["rocprofv3", "--pmc", cname, "--kernel-include-regex", RX, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
                    sys.executable, __file__, steps, "--child"], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rows.sort()
    vals[cname] = [v for _, v in rows]
    subprocess.run(["rm", "-rf", d])
n = len(order)
assert len(vals["FETCH_SIZE"]) == n == len(vals["WRITE_SIZE"]), (n, len(vals["FETCH_SIZE"]), len(vals["WRITE_SIZE"]))
agg = collections.OrderedDict()
for (name, alg), f, w in zip(order, vals["FETCH_SIZE"], vals["WRITE_SIZE"]):
    a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += alg; a[2] += f * 1024 * 2; a[3] += w * 1024          # KB -> bytes, FETCH doubled (gfx950 note)
lines = [f"per-shape HBM traffic of the GEMM kernels, one 25x384x512 clip with {steps} denoise steps; FETCH_SIZE doubled (gfx950), WRITE_SIZE as reported",
         f"{'shape':46s} {'calls':>6s} {'alg MB/call':>12s} {'read MB':>9s} {'write MB':>9s} {'traffic/alg':>11s} {'share of traffic':>16s}"]
tot = sum(a[2] + a[3] for a in agg.values())
for name, a in sorted(agg.items(), key=lambda kv: -(kv[1][2] + kv[1][3])):
    lines.append(f"{name:46s} {a[0]:6d} {a[1] / a[0] / 1e6:12.1f} {a[2] / a[0] / 1e6:9.1f} {a[3] / a[0] / 1e6:9.1f} {(a[2] + a[3]) / max(a[1], 1):11.2f} {100 * (a[2] + a[3]) / tot:15.1f}%")
lines.append(f"TOTAL traffic / algorithmic = {tot / sum(a[1] for a in agg.values()):.2f}")
open("gpurun_out/pmc_per_shape.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:45]))
