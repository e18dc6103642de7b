"""Oracle (test infrastructure): the DepthCrafter pipeline call and the reference wrapper's
post-processing, on CPU.

Follows /root/reference/model/depthcrafter.py:
  * :39-45  ``prepare_input``  (uint8 truncation, /255)
  * :80-90  pipeline kwargs: guidance_scale=1.0 (no CFG, one UNet call per step),
            window_size=len(frames) (=> one window, overlap forced to 0), output_type="np"
  * :92-97  channel mean, clip-global min-max, 1/(x+0.1)
The body of the pipeline call is Tencent/DepthCrafter's ``DepthCrafterPipeline.__call__``
(un-vendored; restated from the public implementation - PARITY UNPINNED).  Noise is an
explicit input because the reference draws it from the global CUDA RNG without a generator
(:80-90), which no other stack can reproduce.
"""
import numpy as np
import torch

from .clip import clip_preprocess
from .scheduler import EulerKarrasVPred

ADDED_TIME_IDS = (7.0, 127.0, 0.02)     # fps (already fps-1... as DepthCrafter passes 7), motion bucket, noise_aug
NOISE_AUG = 0.02
SCALING = 0.18215


def prepare_input(images):
    """model/depthcrafter.py:39-45."""
    frames = [np.asarray(x).transpose(1, 2, 0).astype(np.uint8) for x in images]
    return np.stack(frames, axis=0).astype(np.float32) / 255.0


@torch.no_grad()
def run_pipeline(unet, vae, clip, frames_thwc, noise_latents, noise_aug, steps,
                 chunk=8, dtype=torch.float32, vae_encode_dtype=None, return_stages=False, window=None, overlap=0, timing=None):
    """frames_thwc np/tensor [T,H,W,3] f32 in [0,1]; noise_latents [1,T,4,h,w];
    noise_aug [T,3,H,W] -> np [T,H,W,3] f32 in [0,1].

    window < T switches on the long-video mode of upstream DepthCrafter's pipeline (latent sliding windows; the reference
    never does: model/depthcrafter.py:87-88 passes window_size = len(frames)).  Restated from the published pipeline -
    PARITY UNPINNED: windows advance by window - overlap; a window after the first starts its first `overlap` frames from the
    previous result re-noised to sigma_0; the window's unit noise is the previous one rotated by `overlap` frames; the overlap is
    cross-faded with linspace(0, 1, overlap) into the running result.  The first `window` frames of noise_latents are the noise."""
    import time
    st = {}
    tick = [time.perf_counter()]

    def lap(name):                      # component wall-clock for bench.py's cpu_baseline (timing = dict or None)
        now = time.perf_counter()
        if timing is not None:
            timing[name] = timing.get(name, 0.0) + now - tick[0]
        tick[0] = now
    video = torch.as_tensor(frames_thwc).permute(0, 3, 1, 2).to(dtype)
    video = video * 2.0 - 1.0
    T = video.shape[0]
    # CLIP image embeddings, per frame (DepthCrafter.encode_video)
    emb = []
    for i in range(0, T, chunk):
        emb.append(clip(clip_preprocess(video[i:i + chunk]).to(dtype)))
    emb = torch.cat(emb, 0).unsqueeze(0)                       # [1,T,1024]
    st["clip_emb"] = emb
    lap("clip_s")
    # noise augmentation + VAE encode (fp32 in the reference: force_upcast)
    video = video + NOISE_AUG * noise_aug.to(dtype)
    edt = vae_encode_dtype or torch.float32
    lat = []
    for i in range(0, T, chunk):
        lat.append(vae.to(edt).encode_mode(video[i:i + chunk].to(edt)))
    vae.to(dtype)
    cond = torch.cat(lat, 0).unsqueeze(0).to(dtype)            # [1,T,4,h,w], NOT scaled
    st["cond_latents"] = cond
    lap("vae_encode_s")
    added = torch.tensor([ADDED_TIME_IDS], dtype=dtype)
    sch = EulerKarrasVPred()
    ts = sch.set_timesteps(steps)
    if window is None or window >= T:
        latents = noise_latents.to(dtype) * sch.init_noise_sigma
        for i, t in enumerate(ts):
            x = sch.scale_model_input(latents, i)
            x = torch.cat([x, cond], dim=2)
            v = unet(x, t, emb, added)
            if i == 0:
                st["unet_out0"] = v
            latents = sch.step(v, i, latents)
            st.setdefault("latents_per_step", []).append(latents[0].clone())     # [T,4,h,w] after step i
    else:
        stride = window - overlap
        latents_init = noise_latents[:, :window].to(dtype) * sch.init_noise_sigma
        weights = torch.linspace(0, 1, overlap, dtype=dtype).view(1, overlap, 1, 1, 1) if overlap > 0 else None
        latents_all, idx_start = None, 0
        while idx_start < T - overlap:
            idx_end = min(idx_start + window, T)
            cur = latents_init[:, :idx_end - idx_start].clone()
            latents_init = torch.cat([latents_init[:, -overlap:], latents_init[:, :stride]], dim=1) if overlap > 0 else latents_init
            cond_w, emb_w = cond[:, idx_start:idx_end], emb[:, idx_start:idx_end]
            for i, t in enumerate(ts):
                if latents_all is not None and i == 0 and overlap > 0:
                    cur[:, :overlap] = latents_all[:, -overlap:] + cur[:, :overlap] / sch.init_noise_sigma * sch.sigmas[i]
                x = torch.cat([sch.scale_model_input(cur, i), cond_w], dim=2)
                cur = sch.step(unet(x, t, emb_w, added), i, cur)
            if latents_all is None:
                latents_all = cur.clone()
            else:
                if overlap > 0:
                    latents_all[:, -overlap:] = cur[:, :overlap] * weights + latents_all[:, -overlap:] * (1 - weights)
                latents_all = torch.cat([latents_all, cur[:, overlap:]], dim=1)
            idx_start += stride
        latents = latents_all
    st["latents"] = latents
    lap("unet_s")
    z = latents.flatten(0, 1) / SCALING
    out = []
    for i in range(0, T, chunk):
        zc = z[i:i + chunk]
        out.append(vae.decode(zc, zc.shape[0]))
    fr = torch.cat(out, 0).float()                              # [T,3,H,W]
    fr = (fr / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).contiguous().numpy()
    lap("vae_decode_s")
    return (fr, st) if return_stages else fr


def depth_from_frames(res):
    """model/depthcrafter.py:92-97: [T,H,W,3] -> list of [H,W] depth (np f32)."""
    res = res.sum(-1) / res.shape[-1]
    res = (res - res.min()) / (res.max() - res.min())
    return [1 / (x + 0.1) for x in res]


def stablenormal_post(pred_uint8_list):
    """model/stablenormal.py:40-51: uint8 normal images -> (pred_normals, pred_depths)."""
    ns = [np.array(n) for n in pred_uint8_list]
    for n in ns:
        n[:, :, 0] = -n[:, :, 0]                                  # uint8 wrap: v -> (256-v) % 256
    ns = [n / 255.0 * 2 - 1 for n in ns]
    normals = torch.stack([torch.from_numpy(x).float() for x in ns], 0)
    return normals, torch.zeros_like(normals[..., 0])
