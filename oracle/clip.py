"""Oracle (test infrastructure): CLIP ViT vision tower with projection + the SVD pipeline's
image pre-processing (antialiased resize to 224, CLIP mean/std normalisation).

Reference call site: the ``image_encoder`` / ``feature_extractor`` components of the pipeline
loaded at /root/reference/model/depthcrafter.py:24-29 and exercised inside the call at :80-90
(DepthCrafter's per-frame ``encode_video``).  The tower is restated to match
``transformers.CLIPVisionModelWithProjection`` parameter names; tests/test_oracle_clip.py
checks this file against the real ``transformers`` class (importable here) on random
weights, so THIS part of the oracle is pinned.  The resize helper restates diffusers'
``_resize_with_antialiasing`` (un-vendored; unpinned).
"""
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


@dataclass
class CLIPConfig:
    hidden_size: int = 1280
    intermediate_size: int = 5120
    num_hidden_layers: int = 32
    num_attention_heads: int = 16
    image_size: int = 224
    patch_size: int = 14
    projection_dim: int = 1024
    layer_norm_eps: float = 1e-5


class _Emb(nn.Module):
    def __init__(self, c: CLIPConfig):
        super().__init__()
        n = (c.image_size // c.patch_size) ** 2 + 1
        self.class_embedding = nn.Parameter(torch.randn(c.hidden_size))
        self.patch_embedding = nn.Conv2d(3, c.hidden_size, c.patch_size, c.patch_size, bias=False)
        self.position_embedding = nn.Embedding(n, c.hidden_size)

    def forward(self, x):
        p = self.patch_embedding(x).flatten(2).transpose(1, 2)
        cls = self.class_embedding.expand(x.shape[0], 1, -1)
        return torch.cat([cls, p], dim=1) + self.position_embedding.weight[None]


class _Attn(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.h, self.d = c.num_attention_heads, c.hidden_size // c.num_attention_heads
        self.q_proj = nn.Linear(c.hidden_size, c.hidden_size)
        self.k_proj = nn.Linear(c.hidden_size, c.hidden_size)
        self.v_proj = nn.Linear(c.hidden_size, c.hidden_size)
        self.out_proj = nn.Linear(c.hidden_size, c.hidden_size)

    def forward(self, x):
        b, s, _ = x.shape
        sp = lambda t: t.reshape(b, s, self.h, self.d).transpose(1, 2)
        q, k, v = sp(self.q_proj(x)), sp(self.k_proj(x)), sp(self.v_proj(x))
        w = torch.softmax((q @ k.transpose(-1, -2)) * self.d ** -0.5, dim=-1)
        return self.out_proj((w @ v).transpose(1, 2).reshape(b, s, -1))


class _MLP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.fc1 = nn.Linear(c.hidden_size, c.intermediate_size)
        self.fc2 = nn.Linear(c.intermediate_size, c.hidden_size)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))  # ViT-H (open_clip) uses exact gelu


class _Layer(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layer_norm1 = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.self_attn = _Attn(c)
        self.layer_norm2 = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)
        self.mlp = _MLP(c)

    def forward(self, x):
        x = x + self.self_attn(self.layer_norm1(x))
        return x + self.mlp(self.layer_norm2(x))


class _Encoder(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(c) for _ in range(c.num_hidden_layers)])


class _Vision(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.embeddings = _Emb(c)
        self.pre_layrnorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)  # (sic)
        self.encoder = _Encoder(c)
        self.post_layernorm = nn.LayerNorm(c.hidden_size, eps=c.layer_norm_eps)


class CLIPVisionWithProjection(nn.Module):
    def __init__(self, cfg: CLIPConfig = CLIPConfig()):
        super().__init__()
        self.cfg = cfg
        self.vision_model = _Vision(cfg)
        self.visual_projection = nn.Linear(cfg.hidden_size, cfg.projection_dim, bias=False)

    def forward(self, pixel_values):
        vm = self.vision_model
        x = vm.pre_layrnorm(vm.embeddings(pixel_values))
        for l in vm.encoder.layers:
            x = l(x)
        return self.visual_projection(vm.post_layernorm(x[:, 0]))


def _gauss1d(ks: int, sigma: float) -> torch.Tensor:
    x = torch.arange(ks, dtype=torch.float32) - ks // 2
    if ks % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2 * sigma ** 2))
    return g / g.sum()


def antialias_taps(src: int, dst: int):
    """(kernel size, normalised taps) of the pre-blur for one axis."""
    sigma = max((src / dst - 1.0) / 2.0, 0.001)
    ks = int(max(2.0 * 2 * sigma, 3))
    if ks % 2 == 0:
        ks += 1
    return ks, _gauss1d(ks, sigma)


def resize_with_antialiasing(x: torch.Tensor, size=(224, 224)) -> torch.Tensor:
    """x [N,C,H,W] fp32 -> separable gaussian blur (reflect pad) -> bicubic, align_corners=True."""
    n, c, h, w = x.shape
    ky, gy = antialias_taps(h, size[0])
    kx, gx = antialias_taps(w, size[1])
    xp = F.pad(x, (kx // 2, kx // 2, 0, 0), mode="reflect")
    x = F.conv2d(xp, gx.view(1, 1, 1, kx).expand(c, 1, 1, kx), groups=c)
    xp = F.pad(x, (0, 0, ky // 2, ky // 2), mode="reflect")
    x = F.conv2d(xp, gy.view(1, 1, ky, 1).expand(c, 1, ky, 1), groups=c)
    return F.interpolate(x, size=size, mode="bicubic", align_corners=True)


def clip_preprocess(video_m11: torch.Tensor) -> torch.Tensor:
    """video [T,3,H,W] in [-1,1] (any float dtype) -> CLIP pixel_values fp32 [T,3,224,224]."""
    v = resize_with_antialiasing(video_m11.float(), (224, 224))
    v = (v + 1.0) / 2.0
    mean = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    return (v - mean) / std
