"""Build oracle torch modules from a (numpy, fp16) state dict - test helper."""
import dataclasses

import numpy as np
import torch

from oracle.clip import CLIPConfig, CLIPVisionWithProjection
from oracle.svd_unet import UNetConfig, UNetSpatioTemporal
from oracle.vae import AutoencoderKLTemporalDecoder, VAEConfig


def _load(mod, state):
    sd = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in state.items()}
    missing, unexpected = mod.load_state_dict(sd, strict=True), None
    return mod.eval()


def oracle_unet(cfg, state):
    return _load(UNetSpatioTemporal(UNetConfig(**dataclasses.asdict(cfg))), state)


def oracle_vae(cfg, state):
    return _load(AutoencoderKLTemporalDecoder(VAEConfig(**dataclasses.asdict(cfg))), state)


def oracle_clip(cfg, state):
    return _load(CLIPVisionWithProjection(CLIPConfig(**dataclasses.asdict(cfg))), state)


def oracle_stablenormal(cfgs, states):
    """(SDUNetCfg, VAECfg, DinoCfg) + {component: state} -> the six oracle modules of oracle/stablenormal.py."""
    from oracle.stablenormal import AutoencoderKL, ControlNet, DinoConfig, DinoV2, SDUNet, SDUNetConfig
    u, v, d = cfgs
    uc = SDUNetConfig(**dataclasses.asdict(u)); vc = VAEConfig(**dataclasses.asdict(v)); dc = DinoConfig(**dataclasses.asdict(d))
    return dict(vae=_load(AutoencoderKL(vc), states["vae"]), unet_y=_load(SDUNet(uc), states["unet_yoso"]),
                ctrl_y=_load(ControlNet(uc), states["controlnet_yoso"]), unet_r=_load(SDUNet(uc), states["unet"]),
                ctrl_d=_load(ControlNet(uc, dino_dim=d.hidden_size), states["controlnet_dino"]), dino=_load(DinoV2(dc), states["dino"]))
