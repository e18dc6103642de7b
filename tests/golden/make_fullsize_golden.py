#!/usr/bin/env python
"""Generate tests/golden/fullsize_25step_golden.npz: the CPU oracle's answer for the EXACT workload bench.py times
(BASELINE configs[1]): one 25-frame 384x512 clip, 25 Karras-Euler steps, the real 1.52 B-parameter SVD-UNet / 97.7 M VAE /
ViT-H/14 CLIP with the seeded random weights of ``DepthCrafterPipelineHIP.from_random(seed=42)``, ``synthetic_clip(seed=1234)``
frames and ``make_noise(seed=0)`` noise.

Run in the BUILD container (CPU only, ~1 h on 8 cores, ~30 GB of RAM):
    python tests/golden/make_fullsize_golden.py [--threads N] [--steps 25]
Follows the call /root/reference/model/depthcrafter.py:80-97 through oracle.pipeline.run_pipeline (fp32, float32 VAE encoder)
and the wrapper's post-processing (:92-97); normals by oracle.geometry.prepare_output (:48-69).

What is stored (fp16 where a value is only compared to ~1e-3; ~7 MB in total):
  latents_final      [25,4,48,64] f32   latents after the last Euler step
  latent_absmax / latent_l2 / latent_mean  [25] f64   per-step statistics of the latents after step i
  latents_step       [3,25,4,48,64] f16 latents after steps 1, 13 and 24 (scaled by 1/absmax of that step; scale stored)
  unet_out0_sample   [25,4,48,64] f16   the UNet's first evaluation (sigma = 700), scaled by 1/absmax
  cond_latents       [25,4,48,64] f16   float32 VAE-encoder output (mode, unscaled) - the pipeline casts it to fp16 as well
  clip_emb           [25,1024]   f32
  frames_sub         [25,96,128,3] f16  decoded frames, every 4th pixel
  frames_full        [1,384,512,3] f16  decoded frame 12, every pixel
  depth_sub          [25,96,128] f32    wrapper depth (1/(x+0.1) of the clip-normalised channel mean), every 4th pixel
  frames_min / frames_max               the clip-global extrema the depth normalisation used
  metrics            Abs Rel / delta<1.25 / normal mean / normal median of the oracle's depth + normals against the synthetic
                     ground truth of tests/test_fullsize_golden_gpu.py (so the 3 s.f. check needs no oracle run on the GPU box)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def synthetic_gt(T, H, W):
    """Smooth synthetic ground-truth depth used by the north-star metric check (any fixed GT works: both pipelines are
    scored against the same one)."""
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    return np.stack([2.5 + np.sin(xx / 23.0 + 0.2 * t) * np.cos(yy / 17.0) + 0.004 * yy for t in range(T)], 0).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=max(1, (os.cpu_count() or 2) - 2))
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--frames", type=int, default=25)
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--out", default=None)
    ap.add_argument("--fp16-storage", action="store_true",
                    help="fp16-storage emulation of the reference's fp16 run (round 5): every leaf module of the UNet / CLIP tower / VAE decoder rounds "
                         "its output to fp16 (tests/util.py: fp16_storage), the VAE encoder stays float32 (diffusers force_upcast), and the latents / "
                         "scaled model input / v*c of the Euler step are rounded to fp16 where diffusers' fp16 pipeline rounds them; writes "
                         "fullsize_25step_fp16emu_golden.npz")
    a = ap.parse_args()
    if a.out is None:
        a.out = os.path.join(ROOT, "tests", "golden", "fullsize_25step_fp16emu_golden.npz" if a.fp16_storage else "fullsize_25step_golden.npz")
    import torch
    torch.set_num_threads(a.threads)
    from oracle.geometry import prepare_output
    from oracle.pipeline import depth_from_frames, run_pipeline
    from oracle_build import oracle_clip, oracle_unet, oracle_vae
    from unigeo_amd import weights as W
    from unigeo_amd.harness import depth_evaluation, normal_evaluation
    from unigeo_amd.model.depthcrafter import DepthCrafter
    from unigeo_amd.pipeline import make_noise
    from unigeo_amd.synthetic import synthetic_clip

    T, H, Wd = a.frames, a.height, a.width
    t0 = time.time()
    u, v, c = W.UNetCfg(), W.VAECfg(), W.CLIPCfg()
    su, sv, sc = W.random_state(W.unet_manifest(u), 42), W.random_state(W.vae_manifest(v), 43), W.random_state(W.clip_manifest(c), 44)
    unet, vae, clip = oracle_unet(u, su), oracle_vae(v, sv), oracle_clip(c, sc)
    del su, sv, sc
    print(f"weights built in {time.time() - t0:.0f} s", flush=True)
    sample = synthetic_clip(T, H, Wd, seed=1234)
    frames = DepthCrafter.prepare_input(None, sample)
    nl, na = make_noise(T, H, Wd, seed=0)
    tm = {}
    t0 = time.time()
    if a.fp16_storage:
        import contextlib
        import oracle.pipeline as OP
        from oracle.scheduler import EulerKarrasVPred
        from util import fp16_storage

        class Fp16LatentEuler(EulerKarrasVPred):
            """The scheduler as an fp16 pipeline sees it: fp16 latents in and out, fp16 `v * c` (oracle/scheduler.py: step)."""
            def scale_model_input(self, sample, i):
                return super().scale_model_input(sample.half(), i).half().float()

            def step(self, model_output, i, sample):
                return super().step(model_output.half(), i, sample.half()).float()

        OP.EulerKarrasVPred = Fp16LatentEuler
        ctx = fp16_storage(unet, clip, vae.decoder)
        nl = nl.astype(np.float16).astype(np.float32)
    else:
        import contextlib
        ctx = contextlib.nullcontext()
    with ctx:
        fr, st = run_pipeline(unet, vae, clip, frames, torch.from_numpy(nl), torch.from_numpy(na), steps=a.steps, return_stages=True, timing=tm)
    print(f"oracle pipeline: {time.time() - t0:.0f} s  {tm}", flush=True)
    per = np.stack([x.numpy() for x in st["latents_per_step"]], 0).astype(np.float64)        # [steps,T,4,h,w]
    keep = sorted({min(a.steps - 1, i) for i in (0, a.steps // 2, a.steps - 2)})
    scales = np.array([np.abs(per[i]).max() for i in keep])
    depth = np.stack(depth_from_frames(fr), 0).astype(np.float32)
    K = np.stack(sample["intrinsics"], 0)
    _, n_ref = prepare_output(list(depth), list(K))
    gt_d = synthetic_gt(T, H, Wd)
    _, gt_n = prepare_output(list(gt_d), list(K))
    mask = np.ones((T, H, Wd), bool); mask[:, :3] = False
    md = depth_evaluation(depth, gt_d, custom_mask=mask, align_with_lstsq=True)[0]
    mn = normal_evaluation(n_ref.numpy(), gt_n.numpy(), custom_mask=mask)
    u0 = st["unet_out0"][0].numpy()
    chm = fr.sum(-1) / 3
    np.savez_compressed(
        a.out,
        latents_final=per[-1].astype(np.float32),
        latent_absmax=np.abs(per).max(axis=(1, 2, 3, 4)), latent_l2=np.sqrt((per ** 2).sum(axis=(1, 2, 3, 4))), latent_mean=per.mean(axis=(1, 2, 3, 4)),
        latents_step=np.stack([(per[i] / s) for i, s in zip(keep, scales)], 0).astype(np.float16), latents_step_index=np.array(keep), latents_step_scale=scales,
        unet_out0_sample=(u0 / np.abs(u0).max()).astype(np.float16), unet_out0_scale=np.float64(np.abs(u0).max()),
        cond_latents=st["cond_latents"][0].numpy().astype(np.float16), clip_emb=st["clip_emb"][0].numpy().astype(np.float32),
        frames_sub=fr[:, ::4, ::4].astype(np.float16), frames_full=fr[[T // 2]].astype(np.float16), frames_full_index=np.array([T // 2]),
        depth_sub=depth[:, ::4, ::4].astype(np.float32), frames_min=np.float64(chm.min()), frames_max=np.float64(chm.max()),
        metric_names=np.array(["Abs Rel", "delta < 1.25", "normal mean", "normal median"]),
        metrics=np.array([md["Abs Rel"], md["delta < 1.25"], mn["normal mean"], mn["normal median"]], np.float64),
        geometry=np.array([T, H, Wd, a.steps]), seeds=np.array([42, 1234, 0]),
        fp16_storage=np.int64(1 if a.fp16_storage else 0),
        oracle_seconds=np.array([tm.get(k, 0.0) for k in ("clip_s", "vae_encode_s", "unet_s", "vae_decode_s")]), oracle_threads=np.int64(a.threads))
    print("wrote", a.out, os.path.getsize(a.out), "bytes; metrics", md["Abs Rel"], mn["normal mean"], flush=True)


if __name__ == "__main__":
    main()
