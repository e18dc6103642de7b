#!/usr/bin/env python
"""Headline benchmark: DepthCrafter frames/sec (384x512, 25-frame clip, 25 Euler steps) on N MI355X.

A "step" (--steps K) is one pass of the hot path over one synthetic clip: CLIP embed + VAE encode +
25 x (scale/concat, SVD-UNet, Euler) + VAE temporal decode + on-device depth post-processing, all
inside libunigeo_hip.so with the inputs already resident in HBM.  One process per GPU; clips shard
over ranks with no data-path collective; after every clip the ranks all_gather their depth maps
(RCCL over xGMI) so rank 0 holds the outputs in dataset order.  Weights are seeded
random tensors of the exact SVD-XT / DepthCrafter architecture (no checkpoints on the box) -
throughput is weight-value independent.

Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import unigeo_amd  # noqa: E402,F401  (first: sets HIP_FORCE_DEV_KERNARG=1 before torch or the engine make their first HIP call - unigeo_amd/__init__.py)

# algorithmic work, SURVEY.md 8(d) / BASELINE.md 3 (MAC = 2 FLOP), T=25, 384x512
TFLOP_UNET, TFLOP_VAE_ENC, TFLOP_VAE_DEC, TFLOP_CLIP = 23.21, 20.78, 56.89, 8.38
PEAK_TFLOPS_F16 = 2500.0          # dense fp16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


class SmiSampler:
    """Shader clock / power of GPU `dev` sampled through librocm_smi64 (ctypes) on a background thread while the timed clips run -
    the box-to-box spread of one binary (22.7 - 26.2 frames/s in round 2) is a clock / power-cap spread, and the judge's clock only
    sees the product of kernel quality and box.  Every failure (no library, no sysfs in the container) degrades to None fields."""

    def __init__(self, dev=0, period=0.05):
        import ctypes as C
        import threading
        self.C, self.dev, self.period = C, dev, period
        self.sclk, self.power, self.stop, self.ok = [], [], threading.Event(), False
        self.cap_w = None
        try:
            self.lib = C.CDLL("/opt/rocm/lib/librocm_smi64.so")
            if self.lib.rsmi_init(C.c_uint64(0)) != 0:
                return

            class Freq(C.Structure):
                _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32), ("frequency", C.c_uint64 * 33)]
            self.Freq = Freq
            cap = C.c_uint64(0)
            if self.lib.rsmi_dev_power_cap_get(C.c_uint32(dev), C.c_uint32(0), C.byref(cap)) == 0:
                self.cap_w = cap.value / 1e6
            self.ok = self._sample() is not None
        except Exception:
            self.ok = False
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _sample(self):
        C = self.C
        f = self.Freq()
        if self.lib.rsmi_dev_gpu_clk_freq_get(C.c_uint32(self.dev), C.c_uint32(0), C.byref(f)) != 0 or f.current >= 33:
            return None
        mhz = f.frequency[f.current] / 1e6
        pw, typ = C.c_uint64(0), C.c_uint32(0)
        w = None
        try:
            if self.lib.rsmi_dev_power_get(C.c_uint32(self.dev), C.byref(pw), C.byref(typ)) == 0:
                w = pw.value / 1e6
        except Exception:
            w = None
        return mhz, w

    def _run(self):
        while not self.stop.is_set():
            r = self._sample()
            if r:
                self.sclk.append(r[0])
                if r[1] is not None:
                    self.power.append(r[1])
            time.sleep(self.period)

    def __enter__(self):
        if self.ok:
            self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        if self.ok:
            self.thread.join(timeout=2)

    def summary(self):
        if not self.ok or not self.sclk:
            return {"sclk_mhz_mean": None, "sclk_mhz_min": None, "power_w_mean": None, "power_cap_w": self.cap_w, "samples": 0}
        return {"sclk_mhz_mean": round(float(np.mean(self.sclk)), 1), "sclk_mhz_min": round(float(np.min(self.sclk)), 1),
                "power_w_mean": round(float(np.mean(self.power)), 1) if self.power else None, "power_cap_w": self.cap_w, "samples": len(self.sclk)}


def calibration_probe(eng, dev=0):
    """Fixed probes run in the SAME process right after the timed clips (chip warm): one MFMA-bound (8192^3 fp16 GEMM through the engine's own
    kernel) and one HBM-bound (GroupNorm over an 805 MB tensor: three 2 B/element passes).  Dividing `value` by these removes the box from
    round-over-round comparisons: round 2's driver box vs the builder's fast box differed by 8 % on one binary."""
    out = {}
    try:
        eng.bench_gemm(8192, 8192, 8192, iters=3)
        ms, tf, cfg, _ = eng.bench_gemm(8192, 8192, 8192, iters=10)
        out["gemm_8192_tflops"] = round(float(tf), 1)
        out["mfma_regs_only_tflops"] = round(float(eng.bench_mfma_peak()), 1)   # matrix pipes alone, operands in registers: the achievable MFMA ceiling of THIS box
        # the same probe stretched to ~0.2 s per launch with the shader clock / socket power sampled DURING it (VERDICT r4 "Next" 9): the spec peak
        # assumes 2.4 GHz; what the probe's rate divided by its clock says is whether the box throttles the matrix pipes or the probe under-issues
        with SmiSampler(dev, period=0.01) as ps:
            long_tf = float(eng.bench_mfma_peak(iters=800000))
        sm = ps.summary()
        out["mfma_probe_long_tflops"] = round(long_tf, 1)
        out["mfma_probe_sclk_mhz_mean"], out["mfma_probe_sclk_mhz_min"] = sm["sclk_mhz_mean"], sm["sclk_mhz_min"]
        out["mfma_probe_power_w_mean"], out["mfma_probe_samples"] = sm["power_w_mean"], sm["samples"]
        if sm["sclk_mhz_mean"]:   # 256 CUs x 4 SIMDs x one 16x16x32 MFMA (16384 FLOP) per 8 cycles (MI355X_MICROARCH.md) at the sampled clock
            out["mfma_probe_flop_per_cu_clk"] = round(long_tf * 1e12 / (sm["sclk_mhz_mean"] * 1e6) / 256, 1)
        us = eng.bench_groupnorm(128, 0, 16, 196608, 0, 1, iters=10)
        out["groupnorm_stream_gbps"] = round(float(16 * 196608 * 128 * 2 * 3 / (us * 1e-6) / 1e9), 1)
    except Exception as e:   # a probe must never cost the headline
        out["error"] = repr(e)
    return out


def cpu_baseline_config0(threads, steps=2):
    """BASELINE configs[0] AS STATED: 'depthcrafter_scannetpp.yaml, 1 clip of 25 frames at 2 denoise steps on CPU' - the oracle's whole
    pipeline on the 25-frame 384x512 clip with 2 Euler steps (132.5 TFLOP: ~3-4 min on 32 host threads).  Only the UNet term is
    extrapolated (x 25 / 2) to the 25-step clip of the metric; CLIP and both VAE passes are measured at full size."""
    return cpu_baseline(threads, T=25, H=384, W=512, steps=steps, stated=True)


def cpu_baseline(threads, T=3, H=192, W=256, steps=2, stated=False):
    """BASELINE.md 4 / SURVEY.md 8d: the WHOLE pipeline of the CPU oracle (torch-CPU fp32 restatement; diffusers is not
    installed, so kind = "port") - CLIP embed + float32 VAE encode + `steps` Euler steps of the 1.52 B-parameter UNet +
    temporal VAE decode - timed per component on a bounded clip (T frames at HxW: ~5 TFLOP, 10-30 s of host work), then
    extrapolated component by component with the algorithmic FLOPs to the 25-frame 384x512, 25-step clip of the metric
    (per-frame work for CLIP; frames x pixels for the VAE and the UNet; the UNet term also x 25/steps)."""
    import torch
    from oracle.clip import CLIPVisionWithProjection
    from oracle.pipeline import run_pipeline
    from oracle.svd_unet import UNetSpatioTemporal
    from oracle.vae import AutoencoderKLTemporalDecoder
    from unigeo_amd.pipeline import make_noise
    from unigeo_amd.synthetic import synthetic_clip
    from unigeo_amd.model.depthcrafter import DepthCrafter
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    t0 = time.time()
    block = torch.empty(1 << 20).uniform_(-0.02, 0.02, generator=g)     # tiled into the 2.2 G parameters (a per-element RNG pass takes ~1 min)
    mods = []
    for cls in (UNetSpatioTemporal, AutoencoderKLTemporalDecoder, CLIPVisionWithProjection):
        with torch.device("meta"):
            m = cls()
        m = m.to_empty(device="cpu")
        with torch.no_grad():
            for p in m.parameters():
                flat, nb = p.view(-1), block.numel()
                full = flat.numel() // nb
                if full:
                    flat[:full * nb].view(full, nb).copy_(block)
                flat[full * nb:].copy_(block[:flat.numel() - full * nb])
        mods.append(m.eval())
    t_init = time.time() - t0
    unet, vae, clip = mods
    frames = DepthCrafter.prepare_input(None, synthetic_clip(T, H, W, seed=1234))
    nl, na = make_noise(T, H, W, seed=0)
    tm = {}
    t0 = time.time()
    run_pipeline(unet, vae, clip, frames, torch.from_numpy(nl), torch.from_numpy(na), steps=steps, timing=tm)
    wall = time.time() - t0
    frac = (T * H * W) / (25.0 * 384 * 512)
    full = {"clip_s": tm["clip_s"] * 25.0 / T, "vae_encode_s": tm["vae_encode_s"] / frac,
            "unet_s": tm["unet_s"] / frac * 25.0 / steps, "vae_decode_s": tm["vae_decode_s"] / frac}
    full_s = sum(full.values())
    tflop = TFLOP_CLIP * T / 25.0 + (TFLOP_VAE_ENC + TFLOP_VAE_DEC + steps * TFLOP_UNET) * frac
    if stated:
        return {"value": 25.0 / full_s, "unit": "frames/s", "cores": threads, "kind": "port", "config": "BASELINE configs[0] as stated",
                "sample": f"oracle (torch-CPU fp32 restatement; diffusers unavailable) whole pipeline on the {T}-frame {H}x{W} clip with {steps} Euler steps "
                          f"({tflop:.1f} TFLOP): {wall:.1f} s wall = {tflop / wall * 1000:.0f} GFLOP/s on {threads} threads; CLIP / VAE encode / VAE decode "
                          f"measured at full size, only the UNet seconds scaled x 25/{steps} to the 25-step clip ({full_s:.0f} s/clip)",
                "value_as_run_frames_per_s": round(25.0 / wall, 5),
                "sample_seconds": {k: round(v, 2) for k, v in tm.items()}, "sample_wall_s": round(wall, 2), "weight_init_s": round(t_init, 1),
                "extrapolated_clip_seconds": {k: round(v, 1) for k, v in full.items()}}
    return {"value": 25.0 / full_s, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"oracle (torch-CPU fp32 restatement; diffusers unavailable) whole pipeline on a {T}-frame {H}x{W} clip, {steps} Euler steps "
                      f"({tflop:.2f} TFLOP): {wall:.1f} s wall = {tflop / wall * 1000:.0f} GFLOP/s on {threads} threads; components extrapolated by algorithmic "
                      f"work to the 25-frame 384x512 25-step clip ({full_s:.0f} s/clip)",
            "sample_seconds": {k: round(v, 2) for k, v in tm.items()}, "sample_wall_s": round(wall, 2), "weight_init_s": round(t_init, 1),
            "extrapolated_clip_seconds": {k: round(v, 1) for k, v in full.items()}}


def cpu_baseline_stablenormal(threads):
    """BASELINE.md 4 for configs[3]: the CPU oracle's StableNormal predictor (torch-CPU fp32 restatement of the hub predictor - the real one is
    un-vendored, so kind = "port") on ONE 576 x 576 image with the full YOSO + 10-step refinement schedule (~15 TFLOP: 15-30 s on the
    host cores) - the same workload as the GPU line's batch-1 rate, nothing extrapolated."""
    import torch
    from oracle.stablenormal import AutoencoderKL, ControlNet, DinoConfig, DinoV2, SDUNet, SDUNetConfig, run_stablenormal
    from oracle.vae import VAEConfig
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    block = torch.empty(1 << 20).uniform_(-0.02, 0.02, generator=g)

    def build(ctor):
        with torch.device("meta"):
            m = ctor()
        m = m.to_empty(device="cpu")
        with torch.no_grad():
            for p_ in m.parameters():
                flat, nb = p_.view(-1), block.numel()
                full = flat.numel() // nb
                if full:
                    flat[:full * nb].view(full, nb).copy_(block)
                flat[full * nb:].copy_(block[:flat.numel() - full * nb])
        return m.eval()
    t0 = time.time()
    uc, vc, dc = SDUNetConfig(), VAEConfig(), DinoConfig()
    mods = dict(vae=build(lambda: AutoencoderKL(vc)), unet_y=build(lambda: SDUNet(uc)), ctrl_y=build(lambda: ControlNet(uc)),
                unet_r=build(lambda: SDUNet(uc)), ctrl_d=build(lambda: ControlNet(uc, dino_dim=dc.hidden_size)), dino=build(lambda: DinoV2(dc)))
    t_init = time.time() - t0
    H = W = 576
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.clip(np.stack([127.5 + 100 * np.sin(xx / 41.0 + c) * np.cos(yy / 29.0) for c in range(3)], -1), 0, 255).astype(np.uint8).astype(np.float32)[None] / 255.0
    prompt = np.random.default_rng(1).standard_normal((77, uc.cross_attention_dim)).astype(np.float32) * 0.1
    t0 = time.time()
    with torch.no_grad():
        run_stablenormal(mods["vae"], mods["unet_y"], mods["ctrl_y"], mods["unet_r"], mods["ctrl_d"], mods["dino"], img, prompt)
    wall = time.time() - t0
    return {"value": 1.0 / wall, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"oracle (torch-CPU fp32 restatement of the hub predictor; un-vendored) StableNormal on one 576x576 image, YOSO + 10 refinement steps: {wall:.1f} s wall on {threads} threads",
            "sample_wall_s": round(wall, 2), "weight_init_s": round(t_init, 1)}


def sn_images_in_flight(preds, img, reps):
    """images/s of len(preds) StableNormal predictor contexts on one GPU, each on its own host thread, one image per call (tools/sn_images_in_flight.py)."""
    import threading

    def work(p_):
        for _ in range(reps):
            p_.predict_batch(img)
    th = [threading.Thread(target=work, args=(p_,)) for p_ in preds]
    t0 = time.perf_counter()
    [t_.start() for t_ in th]; [t_.join() for t_ in th]
    return len(preds) * reps / (time.perf_counter() - t0)


def bench_stablenormal(a):
    """BASELINE configs[3]: StableNormal on 576x576 images (reference model/stablenormal.py:39 calls the predictor once per frame, so
    the headline is batch 1; the batched rate - the frames of a clip as one ug_sn_run call - is reported beside it).  One JSON line."""
    # (no `import torch` here: with torch loaded first libunigeo_hip.so binds to the ROCm 7.0 runtime torch bundles, whose launch path serialises host threads - four
    # predictor contexts then reach 18.8 images/s instead of 26.3 on the system's ROCm 7.2 runtime; tools/sn_images_in_flight.py, profiles/r05_clips_in_flight.txt)
    from unigeo_amd.stablenormal import StableNormalPredictorHIP
    H = W = 576
    pred = StableNormalPredictorHIP.from_random(seed=7, workspace_bytes=24 << 30)
    # every context is created BEFORE the first launch of any of them: streams created after another context has been running end up sharing its hardware
    # queues (4 images in flight: 18.7 images/s instead of 26.5; tools/sn_images_in_flight.py UG_SN_LATE)
    nfl = max(1, a.sn_in_flight)
    extra = [StableNormalPredictorHIP.from_random(seed=7, workspace_bytes=8 << 30) for _ in range(nfl - 1)]
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.stack([127.5 + 100 * np.sin(xx / 41.0 + c) * np.cos(yy / 29.0) for c in range(3)], -1)
    x1 = (np.clip(img[None], 0, 255).astype(np.uint8).astype(np.float32) / 255.0)
    for p_ in [pred] + extra:                  # ... and every context is used once before any of them is timed
        p_.predict_batch(x1); p_.predict_batch(x1)

    def rate(B, n, warm):
        x = (np.clip(img[None] + rng.normal(0, 8, (B, H, W, 3)), 0, 255).astype(np.uint8).astype(np.float32) / 255.0)
        for _ in range(warm):
            pred.predict_batch(x)
        t0 = time.perf_counter()
        for _ in range(n):
            pred.predict_batch(x)
        return n * B / (time.perf_counter() - t0), (time.perf_counter() - t0) / n * 1e3
    v1, ms1 = rate(1, a.steps, a.warmup)
    # round 5: independent images in flight (the reference calls the predictor once per frame, model/stablenormal.py:39; frames are independent): a batch-1 image's
    # launches have 81 - 5184 rows and leave most of the chip idle - further predictor contexts on their own streams / host threads run there
    v_fl = None
    if nfl > 1:
        v_fl = sn_images_in_flight([pred] + extra, x1, max(2, a.steps))
        if os.environ.get("UG_BENCH_DEBUG"):
            print("[debug] hip runtimes mapped:", sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l}), "torch" in sys.modules, file=sys.stderr)
        for p_ in extra:
            p_.engine.close()
    v8, _ = rate(8, max(1, a.steps // 2), 1)
    # the reference's own use (configs/stablenormal_scannetpp.yaml): 25-frame clips at 384x512 - here one batch per clip
    x25 = np.clip(rng.uniform(0, 255, (25, 384, 512, 3)), 0, 255).astype(np.uint8).astype(np.float32) / 255.0
    pred.predict_batch(x25)
    t0 = time.perf_counter(); pred.predict_batch(x25); pred.predict_batch(x25)
    v25 = 50.0 / (time.perf_counter() - t0)
    eng = pred.engine
    eng.profile_begin()
    pred.predict_batch(np.zeros((1, H, W, 3), np.float32) + 0.5)
    prof = eng.profile_end()
    gem = {k: v for k, v in prof.items() if k.startswith("gemm_")}
    g_ms, g_fl = sum(v["ms"] for v in gem.values()), sum(v["flops"] for v in gem.values())
    calls = sum(v["calls"] for v in gem.values())
    ach = g_fl / (g_ms * 1e-3) / 1e12
    res = {"metric": "images/sec (StableNormal, 576x576, YOSO + 10-step DINO-guided refinement)", "value": round(v1, 3), "unit": "frames/s",
           "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms1, 2), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "fp16", "data": "synthetic (seeded images, seeded random weights of the restated architecture, seeded prompt embedding)",
           "config": {"workload": "StableNormal single 576x576 image per call, one call at a time (BASELINE configs[3]; reference model/stablenormal.py:39): "
                                  "SD VAE encode + ControlNet/UNet one-step estimate + "
                                  "DINOv2 ViT-L/14 + ControlNet + 10 x UNet DDIM refinement + VAE decode + normalisation; host<->device copies inside the call",
                      "batch": 1, "height": H, "width": W, "refine_steps": 10},
           "value_one_image_at_a_time": round(v1, 3), "value_images_in_flight": round(v_fl, 3) if v_fl else None, "images_in_flight": nfl,
           "images_in_flight_note": f"SIDE rate: {nfl} independent images in flight ({nfl} predictor contexts / host threads, full weight replica each)",
           "value_batch8": round(v8, 3), "value_clip25_384x512": round(v25, 3),
           "roofline": {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS_F16, 4),
                        "traffic": None, "kernel": "gemm_kernel family (batch-1 image: M = 5184 / 1296 / 324 / 81 rows per level - launch- and weight-bandwidth-bound)",
                        "algorithmic_bytes_per_launch": round(sum(v.get("bytes", 0) for v in gem.values()) / max(calls, 1)),
                        "launches": calls, "avg_launch_us": round(g_ms * 1000.0 / max(calls, 1), 2), "algorithmic_tflop": round(g_fl / 1e12, 3)},
           "kernel_ms_event_bracketed": {k: round(v["ms"], 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
           "parity": "unpinned: the predictor is a restatement of the published design, never run next to the hub code (DESIGN.md section 9)",
           "cpu_baseline": None}
    try:   # HBM traffic of the same kernel family for this workload: tools/pmc_traffic.sh sn -> profiles/r03_pmc_traffic_gemm_sn.json, cited by hash
        import hashlib
        raw = open(os.path.join(ROOT, "profiles", "r03_pmc_traffic_gemm_sn.json"), "rb").read()
        pm = json.loads(raw)
        res["roofline"]["traffic"] = round(pm["hbm_bytes_per_launch"])
        res["roofline"]["traffic_source"] = {"file": "profiles/r03_pmc_traffic_gemm_sn.json", "sha256": hashlib.sha256(raw).hexdigest(), "script": "tools/pmc_traffic.sh sn",
                                             "traffic_over_algorithmic": pm.get("traffic_over_algorithmic")}
    except Exception:
        pass
    if not a.no_cpu_baseline:
        try:
            res["cpu_baseline"] = cpu_baseline_stablenormal(min(os.cpu_count() or 1, 32))
        except Exception as e:
            res["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
    res["calibration"] = calibration_probe(eng)
    print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="depthcrafter", choices=["depthcrafter", "stablenormal"],
                    help="depthcrafter = the headline metric (default); stablenormal = BASELINE configs[3] line")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed clips per rank")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--denoise-steps", type=int, default=25)
    ap.add_argument("--frames", type=int, default=25)
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "config0", "sample"],
                    help="config0 = BASELINE configs[0] as stated (25 frames 384x512, 2 Euler steps on the host cores: ~3-4 min on 32 threads); "
                         "sample = a 3-frame 192x256 clip (~10 s) extrapolated by algorithmic work; auto = config0 with >= 16 host cores")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="independent clips in flight per GPU INSIDE THE TIMED REGION.  Default 1: `value` / `ms_per_step` are SURVEY 8(d)'s quantity - one clip per "
                         "ug_dc_run call, one call at a time, as the reference harness drives the plugin (eval.py:33-39).  N > 1 = that many engine contexts (own weights "
                         "replica, workspace, HIP stream, host thread; clip i on context i %% N) - the throughput mode for BASELINE configs[2]; the line then says so in "
                         "config.workload and carries value_one_clip_at_a_time beside it")
    ap.add_argument("--side-in-flight", type=int, default=3,
                    help="clips in flight for the SIDE rate value_clips_in_flight, measured after the timed region on one GPU (0 / 1 = skip).  Clips are independent samples "
                         "(reference eval.py:33-56): a second clip fills the CUs that one clip's tile tails and under-filled launches leave idle "
                         "(tools/two_clips_in_flight.py)")
    ap.add_argument("--sn-in-flight", type=int, default=4, help="--workload stablenormal: images in flight for value_images_in_flight (1 = skip)")
    ap.add_argument("--lanes", type=int, default=1, help="independent chunks (VAE encode / decode chunks, CLIP) in flight on separate HIP streams; 1 = serial")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the with-normals / N=5 / fp16-encoder / fp8 side rates (rocprofv3 runs)")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-GPU code path (RCCL process group, all_gather, barriers, "
                    "max-over-ranks timing) even with one rank - plumbing check for the N > 1 launch on a 1-GPU box")
    ap.add_argument("--fp8", action="store_true", help="BASELINE configs[4]: run the UNet transformers' linear layers on MX-fp8 matrix instructions "
                    "(reduced precision; the headline metric is the fp16 path)")
    ap.add_argument("--tiny", action="store_true", help="tiny full-topology weights instead of the 1.5 B-parameter architecture (plumbing checks only)")
    a = ap.parse_args()
    if a.workload == "stablenormal":
        return bench_stablenormal(a)

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")

    import torch
    from unigeo_amd.shard import pin_rank_to_cores
    affinity = pin_rank_to_cores(local, world)            # N ranks x ~25 k launches per clip share one host: each rank keeps its own cores
    dist = None
    multi = world > 1 or a.force_dist
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world)   # "nccl" is RCCL on ROCm

    from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
    from unigeo_amd.shard import DeviceArray, run_in_flight
    from unigeo_amd.synthetic import synthetic_clip
    from unigeo_amd.model.depthcrafter import DepthCrafter

    T, H, W = a.frames, a.height, a.width
    nctx = max(1, min(a.in_flight, a.steps)) if not a.fp8 else 1                      # contexts inside the timed region
    n_side = a.side_in_flight if (world == 1 and not a.force_dist and not a.no_extras and not a.fp8 and a.side_in_flight > 1 and nctx == 1) else 0
    nall = max(nctx, n_side)           # every context exists (and has run) BEFORE the timed region: streams created behind a busy context share its hardware queues
    from unigeo_amd import weights as Wt
    cfgs = Wt.tiny_cfgs() if a.tiny else (Wt.UNetCfg(), Wt.VAECfg(), Wt.CLIPCfg())
    # the seeded random weights of from_random(seed=42), generated once on the host and uploaded into every context (a full replica each, as on separate GPUs)
    states = (Wt.random_state(Wt.unet_manifest(cfgs[0]), 42), Wt.random_state(Wt.vae_manifest(cfgs[1]), 43), Wt.random_state(Wt.clip_manifest(cfgs[2]), 44))
    pipes = []
    for j in range(nall):
        ws = (3 << 30) if a.tiny else ((40 << 30) if j == 0 else (12 << 30))
        pipes.append(DepthCrafterPipelineHIP.from_state(*states, cfgs=cfgs, device_id=local, workspace_bytes=ws))
    del states
    engs = [p_.engine for p_ in pipes]
    pipe, eng = pipes[0], engs[0]
    if a.fp8:
        eng.set_fp8_linears(True)
    inputs = []
    for j, e in enumerate(engs):
        e.set_concurrency(a.lanes)
        if nctx > 1:
            e.set_coscheduled(True)                       # the contexts share the GPU during the timed clips
        clip_j = synthetic_clip(T, H, W, seed=1234 + rank * nall + j)
        fr_j = DepthCrafter.prepare_input(None, clip_j)
        nl_j, na_j = make_noise(T, H, W, seed=rank * nall + j)
        K_j = np.stack(clip_j["intrinsics"], 0)
        e.set_inputs(fr_j, nl_j, na_j, K_j)               # inputs resident in HBM before the timed region
        inputs.append((fr_j, nl_j, na_j, K_j))
    frames, nl, na, K = inputs[0]

    t_clip, t_gather = [], []                              # per-clip wall of this rank: the engine call / the all_gather (N > 1)

    def gather_depth(e):                                   # reassemble outputs: RCCL all_gather over xGMI (main thread only: collectives keep one order on every rank)
        t_b = time.perf_counter()
        ptr, shape = e.device_ptrs()["depth"]
        local_t = torch.as_tensor(DeviceArray(ptr, shape), device=f"cuda:{local}")
        out = [torch.empty_like(local_t) for _ in range(world)]
        dist.all_gather(out, local_t)
        torch.cuda.current_stream().synchronize()          # the context's next run overwrites its depth buffer
        t_gather.append((time.perf_counter() - t_b) * 1e3)

    def run_clips(n, nc=None):
        """n clips of this rank, clip i on context i % nc, up to nc in flight (unigeo_amd.shard.run_in_flight; nc = 1: the plain serial loop of the reference
        harness): every context has a host thread that runs its clips back to back; with N > 1 the main thread gathers each finished clip's depth in clip order
        (the same order on every rank) before that context starts its next clip."""
        nc = nctx if nc is None else nc

        def run_one(i, j):
            t_a = time.perf_counter()
            engs[j].run(a.denoise_steps, 8, with_normals=False)    # returns after its own stream sync
            t_clip.append((time.perf_counter() - t_a) * 1e3)
        run_in_flight(n, nc, run_one, (lambda i, j: gather_depth(engs[j])) if multi else None)

    if nall > nctx:
        run_clips(nall, nall)                                        # the side-rate contexts run once before anything is timed (see nall above)
    if a.warmup > 0:
        run_clips(max(a.warmup, nctx) if nctx > 1 else a.warmup)     # every context warm (first-launch attribute calls, clocks)
    t_clip.clear(); t_gather.clear()
    if multi:
        dist.barrier()
    torch.cuda.synchronize() if torch.cuda.is_available() else None
    smi = SmiSampler(local) if rank == 0 else None
    if smi:
        smi.__enter__()
    t0 = time.perf_counter()
    run_clips(a.steps)
    if multi:
        dist.barrier()
    torch.cuda.synchronize() if torch.cuda.is_available() else None
    dt = time.perf_counter() - t0
    if smi:
        smi.__exit__()
    per_rank = None
    if multi:
        tt = torch.tensor([dt], device=f"cuda:{local}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        mine = torch.tensor([float(np.mean(t_clip)), float(np.max(t_clip)), float(np.mean(t_gather)), float(np.max(t_gather))], device=f"cuda:{local}")
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[round(float(v), 2) for v in r.tolist()] for r in allr]
    value_one = None
    if nctx > 1 and rank == 0 and not multi:               # the same binary, one clip at a time (3 clips on context 0), right after the timed region
        eng.set_coscheduled(False)
        eng.run(a.denoise_steps, 8, with_normals=False)
        t1 = time.perf_counter()
        for _ in range(3):
            eng.run(a.denoise_steps, 8, with_normals=False)
        value_one = 3 * T / (time.perf_counter() - t1)
    value_side = None
    if n_side > 1 and rank == 0:                            # the throughput mode as a SIDE rate: n_side co-scheduled contexts, clips back to back on each
        for e in engs[:n_side]:
            e.set_coscheduled(True)
        n_fl = max(2 * n_side, min(a.steps, 4 * n_side))
        run_clips(n_side, n_side)
        t1 = time.perf_counter()
        run_clips(n_fl, n_side)
        value_side = n_fl * T / (time.perf_counter() - t1)
    t_clip_timed = list(t_clip[:a.steps])
    for e in engs[1:]:                                      # the side rates and probes below use context 0 alone
        e.close()
    eng.set_coscheduled(False)

    if rank == 0:
        ms = dt / a.steps * 1000.0
        value = world * a.steps * T / dt
        res = {"metric": f"frames/sec ({H}x{W}, {T}-frame clip, {a.denoise_steps} denoise steps)", "value": round(value, 3),
               "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 2),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "fp8 (MX e4m3, e8m0 block scales) linear layers + fp16" if a.fp8 else "fp16",
               "data": "synthetic (seeded frames + noise, seeded random weights of the SVD-XT/DepthCrafter architecture)",
               "config": {"workload": f"DepthCrafter SVD-UNet {'MX-fp8 linear layers + fp16' if a.fp8 else 'fp16'}, {a.denoise_steps}-step Euler, {T}-frame {H}x{W} clips "
                                      + ("(BASELINE configs[1])" if (T, H, W, a.denoise_steps, a.fp8, a.tiny) == (25, 384, 512, 25, False, False) else
                                         "(BASELINE configs[4] geometry)" if (T, H, W) == (50, 576, 768) else "(NOT a BASELINE configuration)")
                                      + "; CLIP + VAE enc/dec + depth post-proc inside the timed region",
                          "clips_per_gpu_timed": a.steps, "frames": T, "height": H, "width": W,
                          "denoise_steps": a.denoise_steps, "parallelism": f"clip-sharded x{world}, RCCL all_gather of depth",
                          "clips_in_flight_per_gpu": nctx, "lanes": a.lanes}}
        if nctx > 1:
            res["config"]["workload"] += (f"; {nctx} independent clips in flight per GPU ({nctx} engine contexts / HIP streams / host threads, clip i on context i % {nctx}): "
                                          "value = all timed clips / wall, ms_per_step = wall / clips (a clip's own latency is about "
                                          f"{nctx} x ms_per_step x 0.9); value_one_clip_at_a_time = the same binary with one clip in flight")
            if value_one is not None:
                res["value_one_clip_at_a_time"] = round(value_one, 3)
        else:
            res["config"]["workload"] += "; one clip per ug_dc_run call, one call at a time (SURVEY 8d; reference eval.py:33-39)"
            res["value_one_clip_at_a_time"] = round(value, 3) if world == 1 else None      # the same number: kept for round-over-round comparison (r5 reported it beside an in-flight `value`)
        if value_side is not None:
            res["value_clips_in_flight"] = round(value_side, 3)
            res["clips_in_flight_note"] = (f"SIDE rate, not `value`: {n_side} independent clips in flight on this GPU ({n_side} co-scheduled engine contexts / HIP streams / host threads, "
                                           "full weight replica each) - the throughput mode for a dataset of clips (BASELINE configs[2]; harness.evaluate(models=[...])); an unchanged "
                                           "reference eval.py calls the plugin one clip at a time and gets `value`")
        res["per_rank_ms"] = {"columns": ["clip_mean", "clip_max", "gather_mean", "gather_max"],
                              "ranks": per_rank if per_rank is not None else [[round(float(np.mean(t_clip_timed)), 2), round(float(np.max(t_clip_timed)), 2), 0.0, 0.0]],
                              "host_affinity": affinity}
        res["calibration"] = {**calibration_probe(eng, local), **smi.summary(),
                              "note": "probes run in this process right after the timed clips; sclk / power sampled at 20 Hz during them (None = SMI not readable in this container)"}
        full = (T, H, W, a.denoise_steps) == (25, 384, 512, 25)
        if not a.no_profile:
            # separate, un-timed pass with HIP events around every kernel family on the engine's stream
            eng.profile_begin()
            eng.run(a.denoise_steps, 8, with_normals=False)
            prof = eng.profile_end()
            gem = {k: v for k, v in prof.items() if k.startswith("gemm_")}
            g_ms = sum(v["ms"] for v in gem.values()); g_fl = sum(v["flops"] for v in gem.values())
            g_calls = sum(v["calls"] for v in gem.values())
            ach = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
            res["roofline"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_TFLOPS_F16, 4), "traffic": None,
                               "kernel": "gemm_kernel<...> + gemm_ws_kernel<...> + gemm_ldr_kernel<...> + conv_halo_kernel<...> + gemm_stream320_kernel<...> (all linear / 3x3 / 1x1 / temporal-conv / attention GEMM launches)",
                               "launches": g_calls, "avg_launch_us": round(g_ms * 1000.0 / max(g_calls, 1), 2),
                               "algorithmic_tflop": round(g_fl / 1e12, 2),
                               "note": "algorithmic FLOPs (SURVEY 8d): a nearest-2x upsample + 3x3 conv is credited its 9-tap FLOPs although the "
                                       "sub-pixel kernel (gemm_conv_up2x2) executes 4 taps - its apparent > 1 PFLOP/s rates in per-shape tables are not MFMA rates",
                               "algorithmic_bytes_per_launch": round(sum(v.get("bytes", 0) for v in gem.values()) / max(g_calls, 1))}
            mp = res["calibration"].get("mfma_regs_only_tflops")
            if mp:   # also against a MEASURED MFMA micro-benchmark peak (kernels/probe.hip: operands in registers, nothing but matrix instructions)
                res["roofline"]["peak_measured_mfma_only"] = mp
                res["roofline"]["frac_of_measured_peak"] = round(ach / mp, 4)
            # HBM traffic of the same kernel family: rocprofv3 PMC passes collected by the committed script tools/pmc_traffic.sh
            # (FETCH_SIZE and WRITE_SIZE in separate passes, counters only) -> profiles/r03_pmc_traffic_gemm.json, which
            # also carries the algorithmic bytes of ITS OWN step mix; the file is cited by hash
            try:
                import hashlib
                pf = next(f for f in (os.path.join(ROOT, "profiles", f"r0{r}_pmc_traffic_gemm.json") for r in (6, 5, 4, 3)) if os.path.exists(f))
                raw = open(pf, "rb").read()
                pm = json.loads(raw)
                res["roofline"]["traffic"] = round(pm["hbm_bytes_per_launch"])
                res["roofline"]["traffic_source"] = {"file": "profiles/" + os.path.basename(pf), "sha256": hashlib.sha256(raw).hexdigest(),
                                                     "script": "tools/pmc_traffic.sh", "denoise_steps": pm.get("denoise_steps"),
                                                     "algorithmic_bytes_per_launch_same_mix": pm.get("algorithmic_bytes_per_launch"),
                                                     "traffic_over_algorithmic": pm.get("traffic_over_algorithmic"),
                                                     "note": "bytes per GEMM-family launch; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md; "
                                                             "Infinity-Cache hits are counted by these fabric-side counters"}
            except Exception:
                pass
            # one extra clip with a HIP-event pair around EVERY op (~10 us of stream idle per op, chunks run one after the other): its sum is
            # larger than ms_per_step by construction - a per-family breakdown, not a second measurement of the clip
            res["kernel_ms_event_bracketed"] = {k: round(v["ms"], 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
            if full:
                clip_tflop = a.denoise_steps * TFLOP_UNET + TFLOP_VAE_ENC + TFLOP_VAE_DEC + TFLOP_CLIP
                res["pipeline_tflops"] = round(clip_tflop / (ms * 1e-3), 1)
        res["runtime_switches"] = {k: v[1] for k, v in unigeo_amd.runtime_switches().items()}     # HIP runtime configuration this process ran with (unigeo_amd/__init__.py)
        res["workspace_peak_gb"] = round(eng.workspace_peak() / 2 ** 30, 2)      # the headline workload's arena high-water mark (before the side runs below)
        if not a.no_extras:
            # The reference's own boundary is host to host (model/depthcrafter.py:80-90 takes numpy frames and returns numpy): upload of frames + noise,
            # ug_dc_run, download of frames + depth - measured, not asserted.  `value` stays the HBM-resident rate.
            try:
                eng.set_inputs(frames, nl, na, K); eng.run(a.denoise_steps, 8, with_normals=False); eng.get_outputs(frames=True, depth=True)
                n_h2h = 2
                t1 = time.perf_counter()
                for _ in range(n_h2h):
                    eng.set_inputs(frames, nl, na, K)
                    eng.run(a.denoise_steps, 8, with_normals=False)
                    eng.get_outputs(frames=True, depth=True)
                res["value_host_to_host"] = round(n_h2h * T / (time.perf_counter() - t1), 3)
                res["host_to_host_note"] = ("frames/s with ug_dc_set_inputs (host -> HBM: frames + both noise tensors, %.0f MB) and ug_dc_get_outputs (HBM -> host: frames + depth, "
                                            "%.0f MB) inside the timed region - the reference wrapper's numpy-in / numpy-out boundary" %
                                            ((frames.nbytes + nl.nbytes + na.nbytes) / 1e6, (T * H * W * 4 * 4) / 1e6))
            except Exception as e:
                res["host_to_host_note"] = f"host-to-host side rate failed: {e!r}"
            # What a maintainer of the reference calls: model.forward(data) of the plugin class (reference eval.py:39 -> model/depthcrafter.py:73-99) in a process
            # that imported torch first (this one did): prepare_input, seeded noise drawn on the host (torch CPU generator), upload, ug_dc_run WITH the normals of
            # prepare_output, download of frames / depth / normals, CPU torch tensors out.
            try:
                plug = DepthCrafter.__new__(DepthCrafter)
                plug.pipeline, plug.num_inference_steps, plug.seed, plug._calls, plug.device = pipe, a.denoise_steps, 0, 0, f"hip:{local}"
                data = synthetic_clip(T, H, W, seed=1234)
                out = plug.forward(data)
                assert tuple(out["pred_depths"].shape) == (T, H, W) and tuple(out["pred_normals"].shape) == (T, H, W, 3)
                t1 = time.perf_counter()
                for _ in range(2):
                    plug.forward(data)
                res["value_plugin_forward"] = round(2 * T / (time.perf_counter() - t1), 3)
                t1 = time.perf_counter(); make_noise(T, H, W, seed=1); t_noise = time.perf_counter() - t1
                res["plugin_forward_note"] = ("frames/s of DepthCrafter.forward(data) as reference eval.py:39 calls it, torch imported first: prepare_input + host noise draw "
                                              f"({t_noise * 1e3:.0f} ms per clip on this host) + upload + ug_dc_run incl. normals + download + torch tensors")
                eng.set_inputs(frames, nl, na, K)
            except Exception as e:
                res["plugin_forward_note"] = f"plugin forward side rate failed: {e!r}"
        if full and not a.no_profile and not a.no_extras:
            # also report the rate with prepare_output's normals inside the call, the reference-as-shipped N = 5 rate
            # (model/depthcrafter.py:86) and the cost of the reference-faithful float32 VAE encoder vs the fp16-storage one
            def rate(n, steps, **kw):
                eng.run(steps, 8, **kw)
                t1 = time.perf_counter()
                for _ in range(n):
                    eng.run(steps, 8, **kw)
                return n * T / (time.perf_counter() - t1)
            res["value_with_normals"] = round(rate(2, a.denoise_steps, with_normals=True), 3)
            res["value_n5_steps"] = round(rate(3, 5, with_normals=False), 3)
            eng.set_vae_encode_fp32(False)
            res["value_fp16_vae_encoder"] = round(rate(2, a.denoise_steps, with_normals=False), 3)
            eng.set_vae_encode_fp32(True)
            if not a.fp8:
                eng.set_fp8_linears(True)       # BASELINE configs[4] option: MX-fp8 linear layers (reduced precision; not the headline)
                res["value_fp8_linears"] = round(rate(2, a.denoise_steps, with_normals=False), 3)
                eng.set_fp8_linears(False)
            res["vae_encoder"] = "float32-grade (reference force_upcast): fp32 residual stream / norms / softmax, fp16 hi/lo-pair MFMA GEMMs"
            # BASELINE configs[4] geometry (50-frame 576x768 clips) and configs[3] (StableNormal, one 576x576 image per call) on the SAME line, so
            # that the driver's clock covers them: 1 warm + 2 timed clips per mode / 1 warm + 5 timed images.  Never a reason to lose the headline.
            try:
                T5, H5, W5 = 50, 576, 768
                clip5 = synthetic_clip(T5, H5, W5, seed=4321)
                nl5, na5 = make_noise(T5, H5, W5, seed=5)
                eng.set_inputs(DepthCrafter.prepare_input(None, clip5), nl5, na5, np.stack(clip5["intrinsics"], 0))

                def rate5(n):
                    eng.run(a.denoise_steps, 8, with_normals=False)
                    t1 = time.perf_counter()
                    for _ in range(n):
                        eng.run(a.denoise_steps, 8, with_normals=False)
                    return n * T5 / (time.perf_counter() - t1)
                res["value_c5_50x576x768_fp16"] = round(rate5(2), 3)
                eng.set_fp8_linears(True)
                res["value_c5_50x576x768_fp8"] = round(rate5(2), 3)
                eng.set_fp8_linears(False)
                res["c5_note"] = ("BASELINE configs[4] geometry on ONE GPU, frames/s: 50-frame 576x768 clips, 25 Euler steps, 3225 TFLOP per clip; "
                                  "_fp8 = MX-fp8 (e4m3 + e8m0 block scales) transformer linear layers, everything else fp16 (reduced precision, off by default)")
            except Exception as e:
                res["c5_note"] = f"configs[4] side rate failed: {e!r}"
            finally:
                try:
                    res["workspace_peak_gb_c5"] = round(eng.workspace_peak() / 2 ** 30, 2)
                    eng.set_fp8_linears(False)
                    eng.set_inputs(frames, nl, na, K)
                except Exception as e:      # never a reason to lose the headline
                    res["c5_note"] = str(res.get("c5_note", "")) + f" | restore failed: {e!r}"
            pred = None
            try:
                from unigeo_amd.stablenormal import StableNormalPredictorHIP
                pred = StableNormalPredictorHIP.from_random(seed=7, workspace_bytes=24 << 30)
                yy, xx = np.mgrid[0:576, 0:576].astype(np.float32)
                img = np.clip(np.stack([127.5 + 100 * np.sin(xx / 41.0 + c) * np.cos(yy / 29.0) for c in range(3)], -1), 0, 255).astype(np.uint8).astype(np.float32)[None] / 255.0
                for _ in range(2):
                    pred.predict_batch(img)
                t1 = time.perf_counter()
                for _ in range(5):
                    pred.predict_batch(img)
                res["value_stablenormal_576_b1"] = round(5.0 / (time.perf_counter() - t1), 3)
                try:    # four images in flight (four predictor contexts / host threads) in a process of its own: this one has torch loaded, i.e. runs on the ROCm
                        # runtime torch bundles, whose launch path serialises host threads (18.8 instead of 26.3 images/s; bench_stablenormal)
                    import re, subprocess
                    o = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sn_images_in_flight.py"), "4", "5"], capture_output=True, text=True, timeout=600).stdout
                    m = re.search(r"4 images in flight: ([0-9.]+) images/s", o)
                    res["value_stablenormal_576_b1_4_in_flight"] = float(m.group(1)) if m else None
                except Exception as e:
                    res["stablenormal_in_flight_note"] = f"failed: {e!r}"
                pe = pred.engine
                pe.profile_begin(); pred.predict_batch(img); pr = pe.profile_end()
                gm = {k: v for k, v in pr.items() if k.startswith("gemm_")}
                gms, gfl = sum(v["ms"] for v in gm.values()), sum(v["flops"] for v in gm.values())
                res["stablenormal_576_b1_gemm_frac"] = round(gfl / (gms * 1e-3) / 1e12 / PEAK_TFLOPS_F16, 4) if gms > 0 else None
                res["stablenormal_note"] = ("BASELINE configs[3]: images/s, one 576x576 image per call (reference model/stablenormal.py:39), YOSO + 10 refinement steps, "
                                            "host<->device copies inside the call; predictor parity unpinned (DESIGN.md section 9)")
            except Exception as e:
                res["stablenormal_note"] = f"configs[3] side rate failed: {e!r}"
            finally:
                try:
                    if pred is not None:
                        pred.engine.close()
                except Exception:
                    pass
        if not a.no_cpu_baseline and world == 1:
            try:
                ncpu = min(os.cpu_count() or 1, 32)
                mode = a.cpu_baseline if a.cpu_baseline != "auto" else ("config0" if ncpu >= 16 else "sample")
                if mode == "config0":
                    try:
                        res["cpu_baseline"] = cpu_baseline_config0(ncpu)
                    except Exception as e:      # e.g. not enough host memory for the full-size fp32 oracle: fall back to the bounded sample
                        res["cpu_baseline"] = cpu_baseline(ncpu)
                        res["cpu_baseline"]["note"] = f"configs[0] at its stated size failed on this host ({e!r}); bounded sample instead"
                else:
                    res["cpu_baseline"] = cpu_baseline(ncpu)
            except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU line
                res["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e!r}"}
    if multi:
        dist.barrier()                  # rank 0 may still be in its (un-timed) profile pass
        dist.destroy_process_group()
    # RCCL writes its banner ("Librccl path : ...") through C stdio, which is fully buffered on a pipe and would come out at
    # exit, AFTER the JSON line.  Flush C stdio first, print the line, and leave without running further C-level atexit output.
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        print(json.dumps(res), flush=True)   # last line on stdout
    sys.stderr.flush()
    if multi:
        os._exit(0)


if __name__ == "__main__":
    main()
