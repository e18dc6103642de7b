"""Oracle (test infrastructure): torch-CPU restatement of the SVD spatio-temporal UNet as
subclassed by DepthCrafter.

Reference call sites: /root/reference/model/depthcrafter.py:15-22 (class + from_pretrained,
fp16) and :80-90 (pipeline call that drives ``unet(...)`` once per denoise step, no CFG).
The algorithm itself is in un-vendored diffusers / Tencent-DepthCrafter code (see
oracle/__init__.py); module and parameter names below follow the diffusers state-dict
layout so that real safetensors files load by name.  PARITY UNPINNED at this boundary.

Layout here is the reference's NCHW ``[B*T, C, H, W]``; the HIP path is channels-last.
"""
import math
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class UNetConfig:
    in_channels: int = 8
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    num_attention_heads: Tuple[int, ...] = (5, 10, 20, 20)
    cross_attention_dim: int = 1024
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 768
    norm_groups: int = 32
    # which down blocks carry transformers (SVD: first three)
    down_has_attn: Tuple[bool, ...] = (True, True, True, False)
    # GroupNorm eps per block family (see SURVEY.md 8(c') uncertainty register)
    eps_cross_attn_blocks: float = 1e-6
    eps_plain_down_block: float = 1e-5
    eps_mid_block: float = 1e-5
    eps_up_blocks: float = 1e-6


def sinusoid(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers ``Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)``."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = t.reshape(-1)[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, hidden, out_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, hidden)
        self.linear_2 = nn.Linear(hidden, out_dim or hidden)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_ch, eps, groups=32):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_ch, cout) if temb_ch else None
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class TemporalResnetBlock(nn.Module):
    """GroupNorm statistics here run over (C/G, T, H, W) of the 5-D tensor, i.e. jointly
    over all frames of the (chunk of the) clip."""

    def __init__(self, cin, cout, temb_ch, eps, groups=32):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv3d(cin, cout, (3, 1, 1), padding=(1, 0, 0))
        self.time_emb_proj = nn.Linear(temb_ch, cout) if temb_ch else None
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv3d(cout, cout, (3, 1, 1), padding=(1, 0, 0))
        assert cin == cout

    def forward(self, x, temb=None):  # x [B,C,T,H,W]; temb [B,T,Ct]
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None:
            t = self.time_emb_proj(F.silu(temb))[:, :, :, None, None].permute(0, 2, 1, 3, 4)
            h = h + t
        h = self.conv2(F.silu(self.norm2(h)))
        return x + h


class AlphaBlender(nn.Module):
    def __init__(self, alpha, strategy="learned_with_images", switch=False):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.tensor([float(alpha)]))
        self.strategy, self.switch = strategy, switch

    def alpha(self):
        # image_only_indicator is all-zero on this path => sigmoid(mix_factor) for both
        # "learned" and "learned_with_images".
        a = torch.sigmoid(self.mix_factor)
        return 1.0 - a if self.switch else a

    def forward(self, x_spatial, x_temporal):
        a = self.alpha().to(x_spatial.dtype)
        return a * x_spatial + (1.0 - a) * x_temporal


class SpatioTemporalResBlock(nn.Module):
    def __init__(self, cin, cout, temb_ch, eps, temporal_eps=None, merge_factor=0.5,
                 strategy="learned_with_images", switch=False, groups=32):
        super().__init__()
        self.spatial_res_block = ResnetBlock2D(cin, cout, temb_ch, eps, groups)
        self.temporal_res_block = TemporalResnetBlock(
            cout, cout, temb_ch, temporal_eps if temporal_eps is not None else eps, groups)
        self.time_mixer = AlphaBlender(merge_factor, strategy, switch)

    def forward(self, x, temb, num_frames):
        x = self.spatial_res_block(x, temb)
        bf, c, h, w = x.shape
        b = bf // num_frames
        x5 = x.reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        t3 = temb.reshape(b, num_frames, -1) if temb is not None else None
        xt = self.temporal_res_block(x5, t3)
        out = self.time_mixer(x5, xt)
        return out.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


class Attention(nn.Module):
    def __init__(self, query_dim, heads, dim_head, cross_dim=None, qkv_bias=False,
                 norm_groups=None, eps=1e-5, residual=False):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head, self.residual = heads, dim_head, residual
        self.group_norm = nn.GroupNorm(norm_groups, query_dim, eps=eps) if norm_groups else None
        self.to_q = nn.Linear(query_dim, inner, bias=qkv_bias)
        self.to_k = nn.Linear(cross_dim or query_dim, inner, bias=qkv_bias)
        self.to_v = nn.Linear(cross_dim or query_dim, inner, bias=qkv_bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim)])

    def forward(self, x, context=None):
        res, shape4 = x, None
        if x.ndim == 4:
            shape4 = x.shape
            x = x.reshape(shape4[0], shape4[1], -1).transpose(1, 2)
        if self.group_norm is not None:
            x = self.group_norm(x.transpose(1, 2)).transpose(1, 2)
        ctx = x if context is None else context
        b, s, _ = x.shape
        q = self.to_q(x).reshape(b, s, self.heads, self.dim_head).transpose(1, 2)
        k = self.to_k(ctx).reshape(b, -1, self.heads, self.dim_head).transpose(1, 2)
        v = self.to_v(ctx).reshape(b, -1, self.heads, self.dim_head).transpose(1, 2)
        w = torch.softmax((q @ k.transpose(-1, -2)) * self.dim_head ** -0.5, dim=-1)
        o = (w @ v).transpose(1, 2).reshape(b, s, -1)
        o = self.to_out[0](o)
        if shape4 is not None:
            o = o.transpose(1, 2).reshape(shape4)
        return o + res if self.residual else o


class GEGLU(nn.Module):
    def __init__(self, din, dout):
        super().__init__()
        self.proj = nn.Linear(din, dout * 2)

    def forward(self, x):
        h, g = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(g)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Identity(),
                                  nn.Linear(dim * mult, dim_out or dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, heads, dim_head, cross_dim=cross_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, context):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context) + x
        return self.ff(self.norm3(x)) + x


class TemporalBasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_dim):
        super().__init__()
        self.norm_in = nn.LayerNorm(dim)
        self.ff_in = FeedForward(dim, dim_out=dim)
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, heads, dim_head, cross_dim=cross_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, num_frames, context):
        bf, s, c = x.shape
        b = bf // num_frames
        x = x.reshape(b, num_frames, s, c).permute(0, 2, 1, 3).reshape(b * s, num_frames, c)
        x = self.ff_in(self.norm_in(x)) + x
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context) + x
        x = self.ff(self.norm3(x)) + x
        return x.reshape(b, s, num_frames, c).permute(0, 2, 1, 3).reshape(bf, s, c)


class TransformerSpatioTemporalModel(nn.Module):
    def __init__(self, heads, dim_head, in_channels, cross_dim, groups=32):
        super().__init__()
        inner = heads * dim_head
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, cross_dim)])
        self.temporal_transformer_blocks = nn.ModuleList(
            [TemporalBasicTransformerBlock(inner, heads, dim_head, cross_dim)])
        self.time_pos_embed = TimestepEmbedding(in_channels, in_channels * 4, in_channels)
        self.time_mixer = AlphaBlender(0.5, "learned_with_images")
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x, context, num_frames):
        bf, c, h, w = x.shape
        b = bf // num_frames
        # temporal cross-attention sees the FIRST frame's context token, for every pixel
        tctx = context.reshape(b, num_frames, -1, context.shape[-1])[:, 0]
        tctx = tctx[:, None].expand(b, h * w, tctx.shape[-2], tctx.shape[-1])
        tctx = tctx.reshape(b * h * w, -1, tctx.shape[-1])
        res = x
        x = self.norm(x).permute(0, 2, 3, 1).reshape(bf, h * w, c)
        x = self.proj_in(x)
        fidx = torch.arange(num_frames).repeat(b)
        emb = self.time_pos_embed(sinusoid(fidx, self.in_channels).to(x.dtype))[:, None, :]
        for blk, tblk in zip(self.transformer_blocks, self.temporal_transformer_blocks):
            x = blk(x, context)
            xm = tblk(x + emb, num_frames, tctx)
            x = self.time_mixer(x, xm)
        x = self.proj_out(x)
        return x.reshape(bf, h, w, c).permute(0, 3, 1, 2) + res


class Downsample2D(nn.Module):
    def __init__(self, ch, padding=1):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:
            x = F.pad(x, (0, 1, 0, 1))
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, layers, heads, cross_dim, has_attn, add_down, eps, g):
        super().__init__()
        self.resnets = nn.ModuleList([
            SpatioTemporalResBlock(cin if i == 0 else cout, cout, temb, eps, groups=g)
            for i in range(layers)])
        self.attentions = nn.ModuleList([
            TransformerSpatioTemporalModel(heads, cout // heads, cout, cross_dim, g)
            for _ in range(layers)]) if has_attn else None
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None

    def forward(self, x, temb, ctx, nf):
        outs = []
        for i, r in enumerate(self.resnets):
            x = r(x, temb, nf)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx, nf)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class MidBlock(nn.Module):
    def __init__(self, ch, temb, heads, cross_dim, eps, g):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlock(ch, ch, temb, eps, groups=g)
                                      for _ in range(2)])
        self.attentions = nn.ModuleList(
            [TransformerSpatioTemporalModel(heads, ch // heads, ch, cross_dim, g)])

    def forward(self, x, temb, ctx, nf):
        x = self.resnets[0](x, temb, nf)
        x = self.attentions[0](x, ctx, nf)
        return self.resnets[1](x, temb, nf)


class UpBlock(nn.Module):
    def __init__(self, cin, cout, prev, temb, layers, heads, cross_dim, has_attn, add_up, eps, g):
        super().__init__()
        rs = []
        for i in range(layers):
            skip = cin if i == layers - 1 else cout
            rin = prev if i == 0 else cout
            rs.append(SpatioTemporalResBlock(rin + skip, cout, temb, eps, groups=g))
        self.resnets = nn.ModuleList(rs)
        self.attentions = nn.ModuleList([
            TransformerSpatioTemporalModel(heads, cout // heads, cout, cross_dim, g)
            for _ in range(layers)]) if has_attn else None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x, skips, temb, ctx, nf):
        for i, r in enumerate(self.resnets):
            x = torch.cat([x, skips.pop()], dim=1)
            x = r(x, temb, nf)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx, nf)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class UNetSpatioTemporal(nn.Module):
    """DepthCrafter variant: per-frame CLIP embeddings ``[B,T,1024] -> [B*T,1,1024]``."""

    def __init__(self, cfg: UNetConfig = UNetConfig()):
        super().__init__()
        self.cfg = cfg
        boc, g = cfg.block_out_channels, cfg.norm_groups
        temb = boc[0] * 4
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, temb)
        n = len(boc)
        downs, ch = [], boc[0]
        for i in range(n):
            has = cfg.down_has_attn[i]
            eps = cfg.eps_cross_attn_blocks if has else cfg.eps_plain_down_block
            downs.append(DownBlock(ch, boc[i], temb, cfg.layers_per_block,
                                   cfg.num_attention_heads[i], cfg.cross_attention_dim,
                                   has, i != n - 1, eps, g))
            ch = boc[i]
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = MidBlock(boc[-1], temb, cfg.num_attention_heads[-1],
                                  cfg.cross_attention_dim, cfg.eps_mid_block, g)
        rev, rheads = list(reversed(boc)), list(reversed(cfg.num_attention_heads))
        rattn = list(reversed(cfg.down_has_attn))
        ups, out_ch = [], rev[0]
        for i in range(n):
            prev, out_ch = out_ch, rev[i]
            cin = rev[min(i + 1, n - 1)]
            ups.append(UpBlock(cin, out_ch, prev, temb, cfg.layers_per_block + 1, rheads[i],
                               cfg.cross_attention_dim, rattn[i], i != n - 1,
                               cfg.eps_up_blocks, g))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-5)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids):
        """sample [B,T,Cin,h,w]; timestep scalar; encoder_hidden_states [B,T,D];
        added_time_ids [B,3] -> [B,T,Cout,h,w]."""
        b, nf = sample.shape[:2]
        dt = self.conv_in.weight.dtype
        t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).expand(b)
        emb = self.time_embedding(sinusoid(t, self.cfg.block_out_channels[0]).to(dt))
        tid = sinusoid(added_time_ids.flatten(), self.cfg.addition_time_embed_dim)
        emb = emb + self.add_embedding(tid.reshape(b, -1).to(dt))
        x = sample.flatten(0, 1).to(dt)
        emb = emb.repeat_interleave(nf, dim=0)
        ctx = encoder_hidden_states.flatten(0, 1).unsqueeze(1).to(dt)
        x = self.conv_in(x)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, emb, ctx, nf)
            skips += outs
        x = self.mid_block(x, emb, ctx, nf)
        for blk in self.up_blocks:
            x = blk(x, skips, emb, ctx, nf)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return x.reshape(b, nf, *x.shape[1:])
