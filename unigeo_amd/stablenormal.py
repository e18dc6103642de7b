"""Host side of the StableNormal predictor on the MI355X-native engine.

Reference interface: ``/root/reference/model/stablenormal.py:16`` builds ``torch.hub.load("Stable-X/StableNormal", "StableNormal",
trust_repo=True)`` and ``:39`` calls it as ``predictor(pil_image) -> pil_normal_image``.  The hub repository is un-vendored and
un-pinned; the object below offers the same call, backed by ``ug_sn_run`` (include/unigeo_hip.h), following the restatement of the
published design in DESIGN.md ("StableNormal uncertainty register", S1-S12).  PARITY UNPINNED at this boundary.

Everything heavy happens inside libunigeo_hip.so.  This file holds the schedule (DDIM coefficients are data handed to the library),
the fixed-prompt text embedding (computed once on the host by ``transformers`` when a checkpoint directory provides the text encoder;
seeded random with synthetic weights) and the uint8 image conversion.
"""
import math
import os

import numpy as np

from . import weights as W
from ._lib import Engine

PROMPT = "The normal map"


def refine_timesteps(start=401, steps=10):
    """`steps` DDIM timesteps from `start` down, evenly spaced: start, start - start/steps, ... (S9)."""
    return [int(round(start - i * start / steps)) for i in range(steps)]


def ddim_tables(timesteps, prediction_type="v_prediction", num_train=1000, beta_start=0.00085, beta_end=0.012):
    """SD scaled-linear betas; deterministic DDIM (eta = 0) from timesteps[i] to timesteps[i+1] (last step: alpha_bar_prev = 1),
    written as the linear update ``x <- a*x + b*model_out`` for epsilon / v / sample prediction (S10)."""
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=np.float64) ** 2
    ab = np.cumprod(1.0 - betas)
    ca, cb = [], []
    for i, t in enumerate(timesteps):
        at = ab[int(t)]
        ap = ab[int(timesteps[i + 1])] if i + 1 < len(timesteps) else 1.0
        sa, s1, pa, p1 = math.sqrt(at), math.sqrt(1 - at), math.sqrt(ap), math.sqrt(1 - ap)
        if prediction_type == "epsilon":
            a, b = pa / sa, p1 - pa * s1 / sa
        elif prediction_type == "v_prediction":
            a, b = pa * sa + p1 * s1, p1 * sa - pa * s1
        elif prediction_type == "sample":
            a, b = p1 / s1, pa - p1 * sa / s1
        else:
            raise ValueError(f"unknown prediction_type {prediction_type!r}")
        ca.append(a); cb.append(b)
    return np.asarray(ca, np.float32), np.asarray(cb, np.float32)


def normals_to_uint8(n):
    """Unit normals in [-1,1] -> the uint8 image the hub predictor returns as PIL: (n + 1) / 2 * 255, truncated (S12)."""
    return (np.clip((np.asarray(n, np.float32) + 1.0) * 0.5, 0.0, 1.0) * 255.0).astype(np.uint8)


COMPONENTS = ("vae", "unet_yoso", "controlnet_yoso", "unet", "controlnet_dino", "dino")


def manifests(cfgs):
    u, v, d = cfgs
    return {"vae": W.sd_vae_manifest(v), "unet_yoso": W.sd_unet_manifest(u), "controlnet_yoso": W.controlnet_manifest(u),
            "unet": W.sd_unet_manifest(u), "controlnet_dino": W.controlnet_manifest(u, d.hidden_size), "dino": W.dino_manifest(d)}


class StableNormalPredictorHIP:
    """Drop-in for the hub ``Predictor``: ``predictor(pil_image) -> pil_normal_image``; ``predict_batch`` takes a whole clip."""

    parity = "unpinned"      # restated from the published design, never run next to the hub predictor

    def __init__(self, engine, cfgs, prompt_embeds, yoso_timestep=999, refine_start=401, refine_steps=10, prediction_type="v_prediction",
                 processing_resolution=0):
        self.engine, self.cfgs = engine, cfgs
        # 0 (default) = process at the input size (S1).  R > 0: the hub predictor's contract as far as it is publicly described - the image is
        # resized so that its longer side is R (both sides rounded to multiples of 64, antialiased bilinear, on the device), processed, and the
        # normals are resized back to the input size and re-normalised.  UNPINNED like the rest of the predictor.
        self.processing_resolution = int(processing_resolution)
        self.prediction_type = prediction_type
        # the DINO tower + DINO ControlNet run on a second HIP stream beside the YOSO estimate (one image's kernels fill a fraction of the chip);
        # bit-identical to the in-order schedule (tests/test_stablenormal_gpu.py)
        try:
            engine.set_concurrency(2)
        except AttributeError:      # an older library selected through UG_LIB_PATH (tools/ab A/B runs) may not export it: in-order schedule
            pass
        self.prompt_embeds = np.ascontiguousarray(prompt_embeds, dtype=np.float32)
        if self.prompt_embeds.shape != (77, cfgs[0].cross_attention_dim):
            raise ValueError(f"prompt_embeds must be [77, {cfgs[0].cross_attention_dim}]")
        self.yoso_timestep = float(yoso_timestep)
        self.timesteps = refine_timesteps(refine_start, refine_steps) if refine_steps > 0 else []
        self.ca, self.cb = ddim_tables(self.timesteps, prediction_type) if self.timesteps else (np.zeros(0, np.float32), np.zeros(0, np.float32))

    @classmethod
    def from_states(cls, states, cfgs=None, prompt_embeds=None, device_id=0, workspace_bytes=None, persist_bytes=None, **kw):
        cfgs = cfgs or (W.SDUNetCfg(), W.VAECfg(), W.DinoCfg())
        nbytes = sum(int(np.prod(a.shape)) * 2 for s in states.values() for a in s.values())
        eng = Engine(device_id, workspace_bytes if workspace_bytes is not None else (24 << 30),
                     persist_bytes if persist_bytes is not None else int(nbytes * 1.2) + (256 << 20))
        for comp in COMPONENTS:
            eng.load_state(f"sn.{comp}.", states[comp])
        eng.bind_stablenormal(*cfgs)
        return cls(eng, cfgs, prompt_embeds, **kw)

    @classmethod
    def from_random(cls, seed=7, cfgs=None, **kw):
        """Seeded synthetic weights of the exact architecture + a seeded stand-in for the text embedding (no checkpoints on the box)."""
        cfgs = cfgs or (W.SDUNetCfg(), W.VAECfg(), W.DinoCfg())
        ms = manifests(cfgs)
        states = {c: W.random_state(ms[c], seed + i) for i, c in enumerate(COMPONENTS)}
        pe = np.random.default_rng(seed + 100).standard_normal((77, cfgs[0].cross_attention_dim)).astype(np.float16).astype(np.float32)
        return cls.from_states(states, cfgs, prompt_embeds=pe, **kw)

    @classmethod
    def from_pretrained(cls, model_dir, cfgs=None, **kw):
        states, pe = W.load_stablenormal_pretrained(model_dir, cfgs)
        return cls.from_states(states, cfgs, prompt_embeds=pe, **kw)

    def predict_batch(self, images01):
        """[B,H,W,3] float in [0,1] -> unit normals [B,H,W,3] float32 in [-1,1] (one ug_sn_run call)."""
        x = np.ascontiguousarray(images01, dtype=np.float32)
        if x.ndim != 4 or x.shape[-1] != 3:
            raise ValueError("images must be [B,H,W,3] float in [0,1]")
        H, Wd = x.shape[1:3]
        R = self.processing_resolution
        if R > 0:
            sc = R / float(max(H, Wd))
            ph, pw = max(64, int(round(H * sc / 64.0)) * 64), max(64, int(round(Wd * sc / 64.0)) * 64)
            if (ph, pw) != (H, Wd):
                n = self.engine.sn_run(self.engine.resize_bilinear(x, ph, pw), self.prompt_embeds, self.yoso_timestep, self.timesteps, self.ca, self.cb)
                return self.engine.resize_bilinear(n, H, Wd, normalise=True)
        if H % 64 or Wd % 64:
            raise ValueError("height and width must be multiples of 64 (or set processing_resolution)")
        return self.engine.sn_run(x, self.prompt_embeds, self.yoso_timestep, self.timesteps, self.ca, self.cb)

    def __call__(self, image):
        """PIL image (or uint8 [H,W,3] array) -> PIL normal image, as ``self.predictor(image)`` at model/stablenormal.py:39."""
        from PIL import Image
        a = np.asarray(image)
        n = self.predict_batch(a[None].astype(np.float32) / 255.0)[0]
        return Image.fromarray(normals_to_uint8(n))
