#!/usr/bin/env python
"""Close the "parity unpinned" gap (SURVEY.md 8c / DESIGN.md 3) on a machine that HAS the reference stack:
diffusers + a Tencent/DepthCrafter checkout + the SVD-XT and DepthCrafter checkpoints (none of which exist in the
build container, so this script is NOT exercised by the test-suite).

It runs the reference pipeline exactly as /root/reference/model/depthcrafter.py:18-34,80-90 constructs and calls it
(fp16, guidance 1.0, window = whole clip) and this repository's HIP pipeline on the same frames with the SAME noise
(the two `randn_tensor` draws of the reference pipeline - augmentation noise, then initial latents - are replaced by
the tensors the HIP path consumes), then reports the differences on decoded frames and on the wrapper's depth.

    python tools/compare_with_diffusers.py --model-dir /path/DepthCrafter --unet-path /path/DepthCrafter-ckpt \
        --pre-train-path /path/stable-video-diffusion-img2vid-xt --steps 5
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-dir", required=True)
    ap.add_argument("--unet-path", required=True)
    ap.add_argument("--pre-train-path", required=True)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--frames", type=int, default=25)
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()

    import torch
    from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
    from unigeo_amd.synthetic import synthetic_clip
    from unigeo_amd.model.depthcrafter import DepthCrafter as HipPlugin

    T, H, W = a.frames, a.height, a.width
    clip = synthetic_clip(T, H, W)
    frames = HipPlugin.prepare_input(None, clip)
    noise_lat, noise_aug = make_noise(T, H, W, a.seed)

    # ---- reference (diffusers) ----
    sys.path.append(a.model_dir)
    from depthcrafter.depth_crafter_ppl import DepthCrafterPipeline
    from depthcrafter.unet import DiffusersUNetSpatioTemporalConditionModelDepthCrafter
    import depthcrafter.depth_crafter_ppl as ppl
    import diffusers.pipelines.stable_video_diffusion.pipeline_stable_video_diffusion as svd

    unet = DiffusersUNetSpatioTemporalConditionModelDepthCrafter.from_pretrained(
        a.unet_path, low_cpu_mem_usage=True, torch_dtype=torch.float16)
    pipe = DepthCrafterPipeline.from_pretrained(a.pre_train_path, unet=unet, torch_dtype=torch.float16, variant="fp16").to("cuda:0")
    queue = [torch.from_numpy(noise_aug), torch.from_numpy(noise_lat)]      # draw order inside the pipeline

    def injected(shape, generator=None, device=None, dtype=None, layout=None):
        t = queue.pop(0)
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
        return t.to(device=device, dtype=dtype)

    ppl.randn_tensor = injected
    svd.randn_tensor = injected
    with torch.inference_mode():
        ref = pipe(frames, height=H, width=W, output_type="np", guidance_scale=1.0, num_inference_steps=a.steps,
                   window_size=T, overlap=25, track_time=False).frames[0]
    del pipe, unet
    torch.cuda.empty_cache()

    # ---- this repository ----
    ours = DepthCrafterPipelineHIP.from_pretrained(a.pre_train_path, a.unet_path)(
        frames, num_inference_steps=a.steps, window_size=T, noise_latents=noise_lat, noise_aug=noise_aug)

    def depth(res):
        d = res.sum(-1) / 3
        d = (d - d.min()) / (d.max() - d.min())
        return 1 / (d + 0.1)

    f_err = np.abs(ours.frames[0] - ref).max()
    d_ref = depth(ref)
    d_rel = np.abs(ours.depth - d_ref).max() / np.abs(d_ref).max()
    abs_rel = float(np.mean(np.abs(ours.depth - d_ref) / d_ref))
    print(f"decoded frames: max |HIP - diffusers| = {f_err:.3e} (frames in [0,1])")
    print(f"wrapper depth : max rel err = {d_rel:.3e}, Abs Rel(HIP vs diffusers) = {abs_rel:.3e}")
    print("north-star bound: 1e-3 relative; see DESIGN.md section 3 for the fp16 noise floor measured against the oracle")


if __name__ == "__main__":
    main()
