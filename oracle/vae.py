"""Oracle (test infrastructure): torch-CPU restatement of ``AutoencoderKLTemporalDecoder``
(SD-VAE encoder + SVD temporal decoder).

Reference call site: the pipeline object built at /root/reference/model/depthcrafter.py:24-31
and driven at :80-90 (VAE encode of the conditioning video, temporal decode of the final
latents in chunks of ``decode_chunk_size``=8).  Algorithm is in un-vendored diffusers;
parameter names follow its state-dict layout.  PARITY UNPINNED at this boundary.
"""
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .svd_unet import (Attention, Downsample2D, ResnetBlock2D, SpatioTemporalResBlock,
                       Upsample2D)


@dataclass
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_groups: int = 32
    scaling_factor: float = 0.18215


class DownEncoderBlock2D(nn.Module):
    def __init__(self, cin, cout, layers, add_down, g):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, None, 1e-6, g)
                                      for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout, padding=0)]) if add_down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class MidBlock2D(nn.Module):
    def __init__(self, ch, g):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, None, 1e-6, g) for _ in range(2)])
        self.attentions = nn.ModuleList([Attention(ch, 1, ch, qkv_bias=True, norm_groups=g,
                                                   eps=1e-6, residual=True)])

    def forward(self, x):
        x = self.resnets[0](x)
        x = self.attentions[0](x)
        return self.resnets[1](x)


class Encoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_groups
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        blocks, ch = [], boc[0]
        for i, c in enumerate(boc):
            blocks.append(DownEncoderBlock2D(ch, c, cfg.layers_per_block, i != len(boc) - 1, g))
            ch = c
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = MidBlock2D(boc[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], 2 * cfg.latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


def _dec_res(cin, cout, g):
    return SpatioTemporalResBlock(cin, cout, None, eps=1e-6, temporal_eps=1e-5, merge_factor=0.0,
                                  strategy="learned", switch=True, groups=g)


class MidBlockTemporalDecoder(nn.Module):
    def __init__(self, ch, layers, g):
        super().__init__()
        self.resnets = nn.ModuleList([_dec_res(ch, ch, g) for _ in range(layers)])
        self.attentions = nn.ModuleList([Attention(ch, 1, ch, qkv_bias=True, norm_groups=g,
                                                   eps=1e-6, residual=True)])

    def forward(self, x, nf):
        x = self.resnets[0](x, None, nf)
        for r, a in zip(self.resnets[1:], self.attentions):
            x = r(a(x), None, nf)
        return x


class UpBlockTemporalDecoder(nn.Module):
    def __init__(self, cin, cout, layers, add_up, g):
        super().__init__()
        self.resnets = nn.ModuleList([_dec_res(cin if i == 0 else cout, cout, g)
                                      for i in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x, nf):
        for r in self.resnets:
            x = r(x, None, nf)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class TemporalDecoder(nn.Module):
    def __init__(self, cfg: VAEConfig):
        super().__init__()
        boc, g = cfg.block_out_channels, cfg.norm_groups
        self.conv_in = nn.Conv2d(cfg.latent_channels, boc[-1], 3, padding=1)
        self.mid_block = MidBlockTemporalDecoder(boc[-1], cfg.layers_per_block, g)
        rev = list(reversed(boc))
        ups, out = [], rev[0]
        for i in range(len(boc)):
            prev, out = out, rev[i]
            ups.append(UpBlockTemporalDecoder(prev, out, cfg.layers_per_block + 1,
                                              i != len(boc) - 1, g))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[0], cfg.out_channels, 3, padding=1)
        self.time_conv_out = nn.Conv3d(cfg.out_channels, cfg.out_channels, (3, 1, 1),
                                       padding=(1, 0, 0))

    def forward(self, z, num_frames):
        x = self.conv_in(z)
        x = self.mid_block(x, num_frames)
        for b in self.up_blocks:
            x = b(x, num_frames)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        bf, c, h, w = x.shape
        b = bf // num_frames
        x = x.reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        x = self.time_conv_out(x)
        return x.permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


class AutoencoderKLTemporalDecoder(nn.Module):
    def __init__(self, cfg: VAEConfig = VAEConfig()):
        super().__init__()
        self.cfg = cfg
        self.encoder = Encoder(cfg)
        self.decoder = TemporalDecoder(cfg)
        self.quant_conv = nn.Conv2d(2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)

    def encode_mode(self, x):
        """``vae.encode(x).latent_dist.mode()`` = mean half of the 2*latent moments."""
        moments = self.quant_conv(self.encoder(x))
        return moments[:, : self.cfg.latent_channels]

    def decode(self, z, num_frames):
        return self.decoder(z, num_frames)
