"""Host-side mirror of the pipeline object the reference plugin drives.

Reference interface (``/root/reference/model/depthcrafter.py``):
  * :18-29  ``DiffusersUNet...from_pretrained(unet_path, torch_dtype=fp16)`` and
            ``DepthCrafterPipeline.from_pretrained(pre_train_path, unet=unet, torch_dtype=fp16, variant="fp16")``
  * :31-34  ``.to(device)``, ``.enable_xformers_memory_efficient_attention()``, ``.enable_attention_slicing()``
  * :80-90  ``pipeline(frames, height=, width=, output_type="np", guidance_scale=1.0,
            num_inference_steps=, window_size=len(frames), overlap=, track_time=False).frames[0]``

Everything heavy happens inside libunigeo_hip.so; this file is argument checking, noise generation
(explicit CPU generator - the reference uses the un-seeded global CUDA RNG) and weight upload.
"""
import dataclasses
from types import SimpleNamespace

import numpy as np

from . import weights as W
from ._lib import Engine


def make_noise(T, H, W_, seed=0):
    """Noise tensors the pipeline consumes, drawn in the order the reference pipeline draws them
    (augmentation noise [T,3,H,W] first, then initial latents [1,T,4,H/8,W/8]) from a CPU
    ``torch.Generator`` so runs are reproducible across stacks."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    aug = torch.randn((T, 3, H, W_), generator=g, dtype=torch.float32)
    lat = torch.randn((1, T, 4, H // 8, W_ // 8), generator=g, dtype=torch.float32)
    return lat.numpy(), aug.numpy()


class DepthCrafterPipelineHIP:
    """Drop-in for ``DepthCrafterPipeline`` as used by the reference wrapper."""

    def __init__(self, engine, unet_cfg, vae_cfg, clip_cfg):
        self.engine, self.unet_cfg, self.vae_cfg, self.clip_cfg = engine, unet_cfg, vae_cfg, clip_cfg
        self.seed = 0
        self.decode_chunk_size = 8

    # ---- construction
    @classmethod
    def from_state(cls, unet_state, vae_state, clip_state, cfgs=None, device_id=0,
                   workspace_bytes=None, persist_bytes=None):
        u, v, c = cfgs or (W.UNetCfg(), W.VAECfg(), W.CLIPCfg())
        nbytes = sum(int(np.prod(a.shape)) * 2 for s in (unet_state, vae_state, clip_state) for a in s.values())
        eng = Engine(device_id,
                     workspace_bytes if workspace_bytes is not None else (24 << 30),
                     persist_bytes if persist_bytes is not None else int(nbytes * 1.35) + (256 << 20))   # weights + re-laid convolution / fused / fp8 copies made at bind time
        eng.load_state("unet.", unet_state); eng.bind_unet(u)
        eng.load_state("vae.", vae_state); eng.bind_vae(v)
        eng.load_state("clip.", clip_state); eng.bind_clip(c)
        return cls(eng, u, v, c)

    @classmethod
    def from_pretrained(cls, pre_train_path, unet_path, device_id=0, **kw):
        """diffusers directory layout (SVD-XT dir + DepthCrafter UNet dir); hard-fails on any tensor
        name/shape that disagrees with the architecture restated by this build."""
        u, v, c, cfgs = W.load_pretrained(unet_path, pre_train_path, with_cfgs=True)    # architecture from the three config.json files
        return cls.from_state(u, v, c, cfgs=cfgs, device_id=device_id, **kw)

    @classmethod
    def from_random(cls, seed=42, cfgs=None, device_id=0, **kw):
        """Seeded synthetic weights of the exact architecture (no checkpoints on the build/bench box)."""
        u, v, c = cfgs or (W.UNetCfg(), W.VAECfg(), W.CLIPCfg())
        return cls.from_state(W.random_state(W.unet_manifest(u), seed),
                              W.random_state(W.vae_manifest(v), seed + 1),
                              W.random_state(W.clip_manifest(c), seed + 2), cfgs=(u, v, c), device_id=device_id, **kw)

    # ---- diffusers-compatible no-ops (reference calls them unconditionally, :31-34)
    def to(self, device):
        return self

    def enable_xformers_memory_efficient_attention(self):
        return None

    def enable_attention_slicing(self):
        return None

    # ---- the call
    def __call__(self, video, height=None, width=None, num_inference_steps=25, guidance_scale=1.0,
                 window_size=110, noise_aug_strength=0.02, decode_chunk_size=None, output_type="np",
                 overlap=25, track_time=False, noise_latents=None, noise_aug=None, seed=None,
                 intrinsics=None, with_normals=False, return_frames=True):
        video = np.asarray(video, dtype=np.float32)
        if video.ndim != 4 or video.shape[-1] != 3:
            raise ValueError("video must be [T,H,W,3] float in [0,1]")
        T, H, Wd, _ = video.shape
        if (height not in (None, H)) or (width not in (None, Wd)):
            raise ValueError("height/width must equal the frame size (the reference passes frames.shape)")
        if H % 64 or Wd % 64:
            raise ValueError("height and width must be multiples of 64")
        if guidance_scale > 1.0:
            raise NotImplementedError("classifier-free guidance is not on the reference path (guidance_scale=1.0)")
        if T > window_size and window_size > 128:
            raise ValueError("a denoising window holds at most 128 frames (temporal attention tile); pass window_size <= 128")
        if T > window_size and not 0 <= overlap < window_size:
            raise ValueError("overlap must be in [0, window_size)")
        if abs(noise_aug_strength - 0.02) > 1e-9:
            raise NotImplementedError("noise_aug_strength is fixed at the pipeline default 0.02")
        if output_type != "np":
            raise NotImplementedError('only output_type="np" (what the reference requests)')
        if noise_latents is None or noise_aug is None:
            noise_latents, noise_aug = make_noise(T, H, Wd, self.seed if seed is None else seed)
        chunk = decode_chunk_size or self.decode_chunk_size
        eng = self.engine
        eng.set_inputs(video, noise_latents, noise_aug, intrinsics)
        # T > window_size: upstream DepthCrafter's latent sliding windows (off on the reference path, which passes window_size = T)
        eng.run(num_inference_steps, chunk, with_normals=with_normals, window=window_size if T > window_size else 0, overlap=overlap)
        # return_frames=False (the plugin's forward, which consumes depth / normals only): the decoded frames stay in HBM - 59 MB less to download per clip
        frames, depth, normals = eng.get_outputs(frames=bool(return_frames), depth=True, normals=with_normals)
        return SimpleNamespace(frames=[frames], depth=depth, normals=normals)
