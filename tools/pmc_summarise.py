#!/usr/bin/env python
"""Sum a PMC counter over the GEMM-family dispatches (gemm_kernel, gemm_ldr_kernel, gemm_ws_kernel) of a rocprofv3 counter_collection CSV and,
given the FETCH_SIZE and WRITE_SIZE pass directories, write the per-launch traffic summary bench.py cites (gpurun_out/pmc_traffic_gemm.json)."""
import csv, glob, hashlib, json, os, sys
out = {}
for d in sys.argv[1:]:
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
    tot, n, name, h = 0.0, 0, None, hashlib.sha256()
    for f in files:
        h.update(open(f, "rb").read())
        for row in csv.DictReader(open(f)):
            kn = row.get("Kernel_Name", "")
            if "gemm_kernel" in kn or "gemm_ldr_kernel" in kn or "gemm_ws_kernel" in kn or "gemm_stream320_kernel" in kn or "ff_fused_kernel" in kn or "conv_halo_kernel" in kn:
                tot += float(row["Counter_Value"]); n += 1; name = row["Counter_Name"]
    out[name or os.path.basename(d)] = {"sum": tot, "dispatch_rows": n, "csv_sha256": h.hexdigest()}
print(json.dumps(out))
json.dump(out, open("gpurun_out/pmc_summary.json", "w"))

if "FETCH_SIZE" in out and "WRITE_SIZE" in out and out["FETCH_SIZE"]["dispatch_rows"]:
    ev = json.load(open("gpurun_out/pmc_events.json")) if os.path.exists("gpurun_out/pmc_events.json") else {}
    n = out["FETCH_SIZE"]["dispatch_rows"]
    # a "launch" is one GEMM launch of the engine (one HIP-event bracket of bench.py's roofline): the row-split 3x3 convolutions are TWO
    # dispatches each, so the divisor is the engine's launch count, not the dispatch count
    nl = ev.get("gemm_launches") or n
    rd = out["FETCH_SIZE"]["sum"] * 1024 * 2 / nl         # counters are in KB; FETCH doubled (gfx950 note)
    wr = out["WRITE_SIZE"]["sum"] * 1024 / nl
    alg = ev.get("algorithmic_bytes_per_launch")
    sn = isinstance(ev.get("denoise_steps"), str)
    json.dump({"workload": ("tools/one_image_sn.py (" + ev["denoise_steps"] + "), GEMM-family dispatches only") if sn else
                           f"tools/one_clip.py {ev.get('denoise_steps')} (25x384x512 clip, CLIP + VAE enc/dec + that many Euler steps), GEMM-family dispatches only",
               "denoise_steps": ev.get("denoise_steps"), "dispatches": n, "hip_event_gemm_launches": ev.get("gemm_launches"),
               "FETCH_SIZE_sum_KB": out["FETCH_SIZE"]["sum"], "WRITE_SIZE_sum_KB": out["WRITE_SIZE"]["sum"],
               "counter_csv_sha256": {k: v["csv_sha256"] for k, v in out.items()},
               "correction": "FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B on wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported",
               "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
               "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (rd + wr) / alg if alg else None,
               "collection": "two separate passes: rocprofv3 --pmc FETCH_SIZE / rocprofv3 --pmc WRITE_SIZE --kernel-include-regex gemm_* (tools/pmc_traffic.sh); fabric-side counters include Infinity-Cache hits"},
              open("gpurun_out/pmc_traffic_gemm.json", "w"), indent=1)
