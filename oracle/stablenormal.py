"""Oracle (test infrastructure): torch-CPU restatement of the StableNormal predictor.

Reference call sites: /root/reference/model/stablenormal.py:16 (``torch.hub.load("Stable-X/StableNormal", "StableNormal")``)
and :39 (``self.predictor(image)`` per frame, PIL in -> PIL normal image out); :40-51 post-processing (pinned by golden G7).

The predictor's code and weights are an un-vendored, un-pinned torch.hub repository (default branch, network only): NOTHING in
/root/reference describes its arithmetic.  This file restates the published design - a one-step "YOSO" normal estimate followed by
a few-step, DINO-guided refinement, both on Stable-Diffusion-2.1-class components - from public knowledge.  PARITY UNPINNED;
every choice below that the reference cannot confirm is listed in DESIGN.md ("StableNormal uncertainty register", S1-S12) and is
enforced at load time by the manifest check of real safetensors (unigeo_amd/weights.py).

Components (state-dict names follow diffusers / dinov2 so real files load by name):
  * ``AutoencoderKL``          SD VAE: the encoder of oracle/vae.py + the plain 2-D decoder + quant / post_quant 1x1 convs
  * ``SDUNet``                 diffusers ``UNet2DConditionModel`` (SD 2.1: 320/640/1280/1280, head_dim 64, linear projections,
                               77-token text cross-attention of width 1024), with ControlNet residual inputs
  * ``ControlNet``             diffusers ``ControlNetModel`` trunk (down + mid blocks, zero 1x1 convs); its ``sample`` is the image
                               latent; the DINO variant adds a projected, nearest-upsampled DINO feature map after ``conv_in``
  * ``DinoV2``                 ViT-L/14 with LayerScale (dinov2 hub naming), 224x224 input, patch tokens out
  * DDIM (eta = 0) coefficients for epsilon / v / sample prediction: every step is ``x <- a*x + b*model_out``
"""
import math
from dataclasses import dataclass
from typing import Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .clip import resize_with_antialiasing
from .svd_unet import (Attention, BasicTransformerBlock, Downsample2D, ResnetBlock2D, TimestepEmbedding, Upsample2D,
                       sinusoid)
from .vae import Encoder, MidBlock2D, VAEConfig

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


@dataclass
class SDUNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    num_attention_heads: Tuple[int, ...] = (5, 10, 20, 20)
    cross_attention_dim: int = 1024
    norm_groups: int = 32
    down_has_attn: Tuple[bool, ...] = (True, True, True, False)


@dataclass
class DinoConfig:
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    image_size: int = 224
    patch_size: int = 14
    layer_norm_eps: float = 1e-6


class Transformer2DModel(nn.Module):
    """GroupNorm(eps 1e-6) -> Linear proj_in -> BasicTransformerBlock (self-attn, 77-token cross-attn, GEGLU FF) -> proj_out + x."""

    def __init__(self, heads, dim_head, ch, cross_dim, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.proj_in = nn.Linear(ch, ch)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(ch, heads, dim_head, cross_dim)])
        self.proj_out = nn.Linear(ch, ch)

    def forward(self, x, ctx):
        b, c, h, w = x.shape
        y = self.norm(x).permute(0, 2, 3, 1).reshape(b, h * w, c)
        y = self.proj_in(y)
        y = self.transformer_blocks[0](y, ctx)
        y = self.proj_out(y)
        return y.reshape(b, h, w, c).permute(0, 3, 1, 2) + x


class _Down(nn.Module):
    def __init__(self, cin, cout, temb, layers, heads, cross, has_attn, add_down, g):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb, 1e-5, g) for i in range(layers)])
        self.attentions = nn.ModuleList([Transformer2DModel(heads, cout // heads, cout, cross, g) for _ in range(layers)]) if has_attn else None
        self.downsamplers = nn.ModuleList([Downsample2D(cout, padding=1)]) if add_down else None

    def forward(self, x, temb, ctx):
        outs = []
        for i, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class _Mid(nn.Module):
    def __init__(self, ch, temb, heads, cross, g):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb, 1e-5, g) for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(heads, ch // heads, ch, cross, g)])

    def forward(self, x, temb, ctx):
        return self.resnets[1](self.attentions[0](self.resnets[0](x, temb), ctx), temb)


class _Up(nn.Module):
    def __init__(self, cins, cout, temb, heads, cross, has_attn, add_up, g):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, cout, temb, 1e-5, g) for c in cins])
        self.attentions = nn.ModuleList([Transformer2DModel(heads, cout // heads, cout, cross, g) for _ in cins]) if has_attn else None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x, skips, temb, ctx):
        for i, r in enumerate(self.resnets):
            x = r(torch.cat([x, skips.pop()], 1), temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class _Trunk(nn.Module):
    """conv_in + time embedding + down blocks + mid block, shared by the UNet and the ControlNet."""

    def __init__(self, cfg: SDUNetConfig):
        super().__init__()
        boc, g, x = cfg.block_out_channels, cfg.norm_groups, cfg.cross_attention_dim
        self.cfg, temb = cfg, boc[0] * 4
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        downs, ch = [], boc[0]
        for i, c in enumerate(boc):
            downs.append(_Down(ch, c, temb, cfg.layers_per_block, cfg.num_attention_heads[i], x, cfg.down_has_attn[i], i != len(boc) - 1, g))
            ch = c
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = _Mid(boc[-1], temb, cfg.num_attention_heads[-1], x, g)

    def trunk(self, sample, t, ctx, after_conv_in=None):
        temb = self.time_embedding(sinusoid(torch.as_tensor(t, dtype=torch.float32).reshape(1), self.cfg.block_out_channels[0]).to(sample.dtype))
        temb = temb.expand(sample.shape[0], -1)
        x = self.conv_in(sample)
        if after_conv_in is not None:
            x = x + after_conv_in
        skips = [x]
        for d in self.down_blocks:
            x, o = d(x, temb, ctx)
            skips += o
        return self.mid_block(x, temb, ctx), skips, temb


class SDUNet(_Trunk):
    def __init__(self, cfg: SDUNetConfig = SDUNetConfig()):
        super().__init__(cfg)
        boc, g, x, temb = cfg.block_out_channels, cfg.norm_groups, cfg.cross_attention_dim, cfg.block_out_channels[0] * 4
        rev, rattn, rheads = list(reversed(boc)), list(reversed(cfg.down_has_attn)), list(reversed(cfg.num_attention_heads))
        ups, out, L, n = [], rev[0], cfg.layers_per_block + 1, len(boc)
        for i in range(n):
            prev, out = out, rev[i]
            cin = rev[min(i + 1, n - 1)]
            cins = [(prev if j == 0 else out) + (cin if j == L - 1 else out) for j in range(L)]
            ups.append(_Up(cins, out, temb, rheads[i], x, rattn[i], i != n - 1, g))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-5)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    def forward(self, sample, t, ctx, down_res=None, mid_res=None):
        """sample [B,4,h,w]; ctx [B,77,1024]; down_res / mid_res: ControlNet residuals added to the skips / the mid output."""
        x, skips, temb = self.trunk(sample, t, ctx)
        if down_res is not None:
            skips = [s + r for s, r in zip(skips, down_res)]
        if mid_res is not None:
            x = x + mid_res
        for u in self.up_blocks:
            x = u(x, skips, temb, ctx)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class ControlNet(_Trunk):
    def __init__(self, cfg: SDUNetConfig = SDUNetConfig(), dino_dim: int = 0):
        super().__init__(cfg)
        boc = cfg.block_out_channels
        chans = [boc[0]]
        for i, c in enumerate(boc):
            chans += [c] * cfg.layers_per_block + ([c] if i != len(boc) - 1 else [])
        self.controlnet_down_blocks = nn.ModuleList([nn.Conv2d(c, c, 1) for c in chans])
        self.controlnet_mid_block = nn.Conv2d(boc[-1], boc[-1], 1)
        self.dino_controlnet_cond_embedding = nn.Linear(dino_dim, boc[0]) if dino_dim else None

    def forward(self, sample, t, ctx, dino_tokens=None, scale=1.0):
        """dino_tokens [B, g*g, D]: projected to conv_in's width, laid out as the g x g grid, nearest-upsampled to the latent grid."""
        add = None
        if self.dino_controlnet_cond_embedding is not None:
            b, n, _ = dino_tokens.shape
            g = int(round(math.sqrt(n)))
            f = self.dino_controlnet_cond_embedding(dino_tokens).reshape(b, g, g, -1).permute(0, 3, 1, 2)
            h, w = sample.shape[-2:]
            iy = (torch.arange(h) * g) // h
            ix = (torch.arange(w) * g) // w
            add = f[:, :, iy][:, :, :, ix]
        x, skips, _ = self.trunk(sample, t, ctx, after_conv_in=add)
        return [z(s) * scale for z, s in zip(self.controlnet_down_blocks, skips)], self.controlnet_mid_block(x) * scale


class Decoder2D(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_groups
        self.conv_in = nn.Conv2d(cfg.latent_channels, boc[-1], 3, padding=1)
        self.mid_block = MidBlock2D(boc[-1], g)
        rev, out, ups = list(reversed(boc)), boc[-1], []
        for i in range(len(boc)):
            prev, out = out, rev[i]
            blk = nn.Module()
            blk.resnets = nn.ModuleList([ResnetBlock2D(prev if j == 0 else out, out, None, 1e-6, g) for j in range(cfg.layers_per_block + 1)])
            blk.upsamplers = nn.ModuleList([Upsample2D(out)]) if i != len(boc) - 1 else None
            ups.append(blk)
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            for r in b.resnets:
                x = r(x)
            if b.upsamplers is not None:
                x = b.upsamplers[0](x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKL(nn.Module):
    def __init__(self, cfg: VAEConfig = VAEConfig()):
        super().__init__()
        self.cfg = cfg
        self.encoder = Encoder(cfg)
        self.decoder = Decoder2D(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(cfg.latent_channels, cfg.latent_channels, 1)

    def encode_mode(self, x):
        return self.quant_conv(self.encoder(x))[:, : self.cfg.latent_channels]

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))


class _DinoBlock(nn.Module):
    def __init__(self, c: DinoConfig):
        super().__init__()
        d = c.hidden_size
        self.norm1 = nn.LayerNorm(d, eps=c.layer_norm_eps)
        self.attn = nn.Module()
        self.attn.qkv = nn.Linear(d, 3 * d)
        self.attn.proj = nn.Linear(d, d)
        self.ls1 = nn.Module(); self.ls1.gamma = nn.Parameter(torch.ones(d))
        self.norm2 = nn.LayerNorm(d, eps=c.layer_norm_eps)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(d, c.intermediate_size)
        self.mlp.fc2 = nn.Linear(c.intermediate_size, d)
        self.ls2 = nn.Module(); self.ls2.gamma = nn.Parameter(torch.ones(d))
        self.heads = c.num_attention_heads

    def forward(self, x):
        b, s, d = x.shape
        q, k, v = self.attn.qkv(self.norm1(x)).reshape(b, s, 3, self.heads, d // self.heads).permute(2, 0, 3, 1, 4)
        w = torch.softmax((q @ k.transpose(-1, -2)) * (d // self.heads) ** -0.5, dim=-1)
        x = x + self.ls1.gamma * self.attn.proj((w @ v).transpose(1, 2).reshape(b, s, d))
        return x + self.ls2.gamma * self.mlp.fc2(F.gelu(self.mlp.fc1(self.norm2(x))))


class DinoV2(nn.Module):
    def __init__(self, cfg: DinoConfig = DinoConfig()):
        super().__init__()
        self.cfg, d, g = cfg, cfg.hidden_size, cfg.image_size // cfg.patch_size
        self.cls_token = nn.Parameter(torch.zeros(1, 1, d))
        self.pos_embed = nn.Parameter(torch.zeros(1, 1 + g * g, d))
        self.patch_embed = nn.Module()
        self.patch_embed.proj = nn.Conv2d(3, d, cfg.patch_size, stride=cfg.patch_size)
        self.blocks = nn.ModuleList([_DinoBlock(cfg) for _ in range(cfg.num_hidden_layers)])
        self.norm = nn.LayerNorm(d, eps=cfg.layer_norm_eps)

    def forward(self, pixel_values):
        """[B,3,S,S] normalised -> patch tokens [B, g*g, D] (x_norm_patchtokens)."""
        x = self.patch_embed.proj(pixel_values).flatten(2).transpose(1, 2)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], 1) + self.pos_embed
        for b in self.blocks:
            x = b(x)
        return self.norm(x)[:, 1:]


def dino_preprocess(img_m11: torch.Tensor, size: int = 224) -> torch.Tensor:
    """[-1,1] image -> antialiased resize to size x size (the resampler of oracle/clip.py) -> [0,1] -> ImageNet mean/std."""
    v = (resize_with_antialiasing(img_m11, (size, size)) + 1.0) / 2.0
    return (v - torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)) / torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)


def ddim_tables(timesteps, prediction_type="v_prediction", num_train=1000, beta_start=0.00085, beta_end=0.012):
    """SD scaled-linear betas; DDIM eta = 0 from timesteps[i] to timesteps[i+1] (final: alpha_bar_prev = 1) as x <- a x + b out."""
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
            raise ValueError(prediction_type)
        ca.append(a); cb.append(b)
    return np.asarray(ca, np.float32), np.asarray(cb, np.float32)


def refine_timesteps(start=401, steps=10):
    """`steps` DDIM timesteps from `start` down, evenly spaced ("trailing"-style): start, start - start/steps, ..."""
    return [int(round(start - i * start / steps)) for i in range(steps)]


@torch.no_grad()
def run_stablenormal(vae, unet_y, ctrl_y, unet_r, ctrl_d, dino, images_bhwc, prompt_embeds, yoso_t=999, refine_start=401,
                     refine_steps=10, prediction_type="v_prediction", dtype=torch.float32, return_stages=False):
    """images [B,H,W,3] in [0,1] (H, W multiples of 64), prompt_embeds [77,1024] -> unit normals [B,H,W,3] in [-1,1].

    1. z_img = 0.18215 * mode(vae.encode(2x-1))
    2. YOSO (one step): residuals = ctrl_y(z_img, t=yoso_t); x = unet_y(zeros, yoso_t, residuals)      (sample prediction)
    3. refinement: residuals = ctrl_d(z_img, t=0, DINO tokens) ONCE; for t in timesteps: x <- a x + b unet_r(x, t, residuals)
    4. decode x / 0.18215, clip to [-1,1], L2-normalise per pixel."""
    st = {}
    img = torch.as_tensor(images_bhwc).permute(0, 3, 1, 2).to(dtype) * 2.0 - 1.0
    B = img.shape[0]
    ctx = torch.as_tensor(prompt_embeds).to(dtype)[None].expand(B, -1, -1)
    z_img = vae.encode_mode(img) * vae.cfg.scaling_factor
    st["z_img"] = z_img
    dr, mr = ctrl_y(z_img, yoso_t, ctx)
    x = unet_y(torch.zeros_like(z_img), yoso_t, ctx, dr, mr)
    st["yoso_latent"] = x
    tok = dino(dino_preprocess(img, dino.cfg.image_size).to(dtype))
    st["dino_tokens"] = tok
    dr, mr = ctrl_d(z_img, 0, ctx, dino_tokens=tok)
    ts = refine_timesteps(refine_start, refine_steps)
    ca, cb = ddim_tables(ts, prediction_type)
    for i, t in enumerate(ts):
        out = unet_r(x, t, ctx, dr, mr)
        x = float(ca[i]) * x + float(cb[i]) * out
    st["latent"] = x
    n = vae.decode(x / vae.cfg.scaling_factor).clamp(-1, 1)
    n = n / n.norm(dim=1, keepdim=True).clamp_min(1e-6)
    out = n.permute(0, 2, 3, 1).contiguous().float().numpy()
    return (out, st) if return_stages else out


def normals_to_uint8(n):
    """What the hub predictor hands back as a PIL image: (n + 1) / 2 * 255, truncated to uint8."""
    return (np.clip((np.asarray(n, np.float32) + 1.0) * 0.5, 0.0, 1.0) * 255.0).astype(np.uint8)
