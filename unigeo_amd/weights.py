"""Weight manifests and loaders for the DepthCrafter components.

The manifests enumerate every tensor (diffusers / transformers state-dict name -> shape) the
HIP engine binds.  They are the enforcement point for the "uncertainty register" of
SURVEY.md 8(c'): loading real safetensors hard-fails on any missing, unexpected or
mis-shaped tensor, so a wrong architectural restatement cannot load silently.

Reference: weights are loaded by ``from_pretrained`` at /root/reference/model/depthcrafter.py:18-29
(``unet_path`` = DepthCrafter UNet dir, ``pre_train_path`` = stable-video-diffusion-img2vid-xt dir,
fp16 variant).
"""
import json
import os
from collections import OrderedDict
from dataclasses import dataclass, field, asdict
from typing import Dict, Tuple

import numpy as np


@dataclass
class UNetCfg:
    in_channels: int = 8
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    num_attention_heads: Tuple[int, ...] = (5, 10, 20, 20)
    cross_attention_dim: int = 1024
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 768
    norm_groups: int = 32
    down_has_attn: Tuple[bool, ...] = (True, True, True, False)
    eps_cross_attn_blocks: float = 1e-6
    eps_plain_down_block: float = 1e-5
    eps_mid_block: float = 1e-5
    eps_up_blocks: float = 1e-6


@dataclass
class VAECfg:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_groups: int = 32
    scaling_factor: float = 0.18215


@dataclass
class CLIPCfg:
    hidden_size: int = 1280
    intermediate_size: int = 5120
    num_hidden_layers: int = 32
    num_attention_heads: int = 16
    image_size: int = 224
    patch_size: int = 14
    projection_dim: int = 1024
    layer_norm_eps: float = 1e-5


def tiny_cfgs():
    """Small configuration with the full topology, for parity tests that the CPU oracle
    finishes in seconds."""
    u = UNetCfg(block_out_channels=(64, 128, 128, 128), num_attention_heads=(1, 2, 2, 2),
                cross_attention_dim=48, addition_time_embed_dim=16,
                projection_class_embeddings_input_dim=48, norm_groups=16)
    v = VAECfg(block_out_channels=(32, 64, 64, 64), norm_groups=8)
    c = CLIPCfg(hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                num_attention_heads=2, image_size=224, patch_size=14, projection_dim=48)
    return u, v, c


class _M(OrderedDict):
    def lin(self, p, cin, cout, bias=True):
        self[p + ".weight"] = (cout, cin)
        if bias:
            self[p + ".bias"] = (cout,)

    def norm(self, p, c):
        self[p + ".weight"] = (c,)
        self[p + ".bias"] = (c,)

    def conv2(self, p, cin, cout, k=3, bias=True):
        self[p + ".weight"] = (cout, cin, k, k)
        if bias:
            self[p + ".bias"] = (cout,)

    def conv3(self, p, cin, cout):
        self[p + ".weight"] = (cout, cin, 3, 1, 1)
        self[p + ".bias"] = (cout,)


def _st_res(m, p, cin, cout, temb):
    s, t = p + ".spatial_res_block", p + ".temporal_res_block"
    m.norm(s + ".norm1", cin); m.conv2(s + ".conv1", cin, cout)
    if temb:
        m.lin(s + ".time_emb_proj", temb, cout)
    m.norm(s + ".norm2", cout); m.conv2(s + ".conv2", cout, cout)
    if cin != cout:
        m.conv2(s + ".conv_shortcut", cin, cout, k=1)
    m.norm(t + ".norm1", cout); m.conv3(t + ".conv1", cout, cout)
    if temb:
        m.lin(t + ".time_emb_proj", temb, cout)
    m.norm(t + ".norm2", cout); m.conv3(t + ".conv2", cout, cout)
    m[p + ".time_mixer.mix_factor"] = (1,)


def _attn(m, p, dim, inner, cross=None, bias=False):
    m.lin(p + ".to_q", dim, inner, bias)
    m.lin(p + ".to_k", cross or dim, inner, bias)
    m.lin(p + ".to_v", cross or dim, inner, bias)
    m.lin(p + ".to_out.0", inner, dim, True)


def _ff(m, p, dim, dout=None):
    m.lin(p + ".net.0.proj", dim, dim * 8)
    m.lin(p + ".net.2", dim * 4, dout or dim)


def _transformer(m, p, ch, cross):
    m.norm(p + ".norm", ch); m.lin(p + ".proj_in", ch, ch)
    b = p + ".transformer_blocks.0"
    m.norm(b + ".norm1", ch); _attn(m, b + ".attn1", ch, ch)
    m.norm(b + ".norm2", ch); _attn(m, b + ".attn2", ch, ch, cross)
    m.norm(b + ".norm3", ch); _ff(m, b + ".ff", ch)
    t = p + ".temporal_transformer_blocks.0"
    m.norm(t + ".norm_in", ch); _ff(m, t + ".ff_in", ch)
    m.norm(t + ".norm1", ch); _attn(m, t + ".attn1", ch, ch)
    m.norm(t + ".norm2", ch); _attn(m, t + ".attn2", ch, ch, cross)
    m.norm(t + ".norm3", ch); _ff(m, t + ".ff", ch)
    m.lin(p + ".time_pos_embed.linear_1", ch, ch * 4)
    m.lin(p + ".time_pos_embed.linear_2", ch * 4, ch)
    m[p + ".time_mixer.mix_factor"] = (1,)
    m.lin(p + ".proj_out", ch, ch)


def unet_manifest(cfg: UNetCfg = UNetCfg()) -> "OrderedDict[str, tuple]":
    m = _M()
    boc = cfg.block_out_channels
    n, temb, x = len(boc), boc[0] * 4, cfg.cross_attention_dim
    m.conv2("conv_in", cfg.in_channels, boc[0])
    m.lin("time_embedding.linear_1", boc[0], temb); m.lin("time_embedding.linear_2", temb, temb)
    m.lin("add_embedding.linear_1", cfg.projection_class_embeddings_input_dim, temb)
    m.lin("add_embedding.linear_2", temb, temb)
    ch = boc[0]
    for i in range(n):
        for j in range(cfg.layers_per_block):
            _st_res(m, f"down_blocks.{i}.resnets.{j}", ch if j == 0 else boc[i], boc[i], temb)
            if cfg.down_has_attn[i]:
                _transformer(m, f"down_blocks.{i}.attentions.{j}", boc[i], x)
        if i != n - 1:
            m.conv2(f"down_blocks.{i}.downsamplers.0.conv", boc[i], boc[i])
        ch = boc[i]
    _st_res(m, "mid_block.resnets.0", boc[-1], boc[-1], temb)
    _transformer(m, "mid_block.attentions.0", boc[-1], x)
    _st_res(m, "mid_block.resnets.1", boc[-1], boc[-1], temb)
    rev, rattn = list(reversed(boc)), list(reversed(cfg.down_has_attn))
    out = rev[0]
    L = cfg.layers_per_block + 1
    for i in range(n):
        prev, out = out, rev[i]
        cin = rev[min(i + 1, n - 1)]
        for j in range(L):
            skip = cin if j == L - 1 else out
            rin = prev if j == 0 else out
            _st_res(m, f"up_blocks.{i}.resnets.{j}", rin + skip, out, temb)
            if rattn[i]:
                _transformer(m, f"up_blocks.{i}.attentions.{j}", out, x)
        if i != n - 1:
            m.conv2(f"up_blocks.{i}.upsamplers.0.conv", out, out)
    m.norm("conv_norm_out", boc[0]); m.conv2("conv_out", boc[0], cfg.out_channels)
    return m


def _res2d(m, p, cin, cout):
    m.norm(p + ".norm1", cin); m.conv2(p + ".conv1", cin, cout)
    m.norm(p + ".norm2", cout); m.conv2(p + ".conv2", cout, cout)
    if cin != cout:
        m.conv2(p + ".conv_shortcut", cin, cout, k=1)


def _vae_attn(m, p, ch):
    m.norm(p + ".group_norm", ch)
    _attn(m, p, ch, ch, bias=True)


def vae_manifest(cfg: VAECfg = VAECfg()) -> "OrderedDict[str, tuple]":
    m = _M()
    boc, L = cfg.block_out_channels, cfg.layers_per_block
    m.conv2("encoder.conv_in", cfg.in_channels, boc[0])
    ch = boc[0]
    for i, c in enumerate(boc):
        for j in range(L):
            _res2d(m, f"encoder.down_blocks.{i}.resnets.{j}", ch if j == 0 else c, c)
        if i != len(boc) - 1:
            m.conv2(f"encoder.down_blocks.{i}.downsamplers.0.conv", c, c)
        ch = c
    _res2d(m, "encoder.mid_block.resnets.0", ch, ch)
    _vae_attn(m, "encoder.mid_block.attentions.0", ch)
    _res2d(m, "encoder.mid_block.resnets.1", ch, ch)
    m.norm("encoder.conv_norm_out", ch); m.conv2("encoder.conv_out", ch, 2 * cfg.latent_channels)
    m.conv2("quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, k=1)
    m.conv2("decoder.conv_in", cfg.latent_channels, boc[-1])
    for j in range(L):
        _st_res(m, f"decoder.mid_block.resnets.{j}", boc[-1], boc[-1], None)
    _vae_attn(m, "decoder.mid_block.attentions.0", boc[-1])
    rev, out = list(reversed(boc)), boc[-1]
    for i in range(len(boc)):
        prev, out = out, rev[i]
        for j in range(L + 1):
            _st_res(m, f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else out, out, None)
        if i != len(boc) - 1:
            m.conv2(f"decoder.up_blocks.{i}.upsamplers.0.conv", out, out)
    m.norm("decoder.conv_norm_out", boc[0]); m.conv2("decoder.conv_out", boc[0], cfg.out_channels)
    m.conv3("decoder.time_conv_out", cfg.out_channels, cfg.out_channels)
    return m


def clip_manifest(cfg: CLIPCfg = CLIPCfg()) -> "OrderedDict[str, tuple]":
    m = _M()
    d, v = cfg.hidden_size, "vision_model"
    ntok = (cfg.image_size // cfg.patch_size) ** 2 + 1
    m[v + ".embeddings.class_embedding"] = (d,)
    m[v + ".embeddings.patch_embedding.weight"] = (d, 3, cfg.patch_size, cfg.patch_size)
    m[v + ".embeddings.position_embedding.weight"] = (ntok, d)
    m.norm(v + ".pre_layrnorm", d)
    for i in range(cfg.num_hidden_layers):
        p = f"{v}.encoder.layers.{i}"
        m.norm(p + ".layer_norm1", d)
        for nme in ("q_proj", "k_proj", "v_proj", "out_proj"):
            m.lin(f"{p}.self_attn.{nme}", d, d)
        m.norm(p + ".layer_norm2", d)
        m.lin(p + ".mlp.fc1", d, cfg.intermediate_size); m.lin(p + ".mlp.fc2", cfg.intermediate_size, d)
    m.norm(v + ".post_layernorm", d)
    m[("visual_projection.weight")] = (cfg.projection_dim, d)
    return m


def random_state(manifest, seed: int, dtype=np.float16) -> Dict[str, np.ndarray]:
    """Seeded synthetic weights (there are no checkpoints on the build / bench boxes).
    Linear/conv weights ~ N(0, 1/fan_in) (keeps activations O(1) through the depth so fp16
    parity tests are meaningful), norm gains ~ 1 + 0.1 N, biases ~ 0.05 N, mixers ~ 0.5 N."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in manifest.items():
        if name.endswith("mix_factor"):
            a = rng.normal(0, 0.5, shape)
        elif name.endswith(".gamma"):                       # DINOv2 LayerScale
            a = 0.5 + 0.1 * rng.standard_normal(shape)
        elif name in ("cls_token", "pos_embed"):
            a = (0.5 if name == "cls_token" else 0.1) * rng.standard_normal(shape)
        elif len(shape) == 1:
            is_gain = name.endswith(".weight")
            a = (1.0 + 0.1 * rng.standard_normal(shape)) if is_gain else 0.05 * rng.standard_normal(shape)
            if name.endswith("class_embedding"):
                a = 0.5 * rng.standard_normal(shape)
        elif "position_embedding" in name:
            a = 0.1 * rng.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            n = int(np.prod(shape))
            if n > (1 << 22):   # big tensors: tile a 4M-sample block (keeps full-size weight synthesis to seconds)
                a = np.resize(rng.standard_normal(1 << 22, dtype=np.float32), n).reshape(shape)
            else:
                a = rng.standard_normal(shape, dtype=np.float32)
            a = a * np.float32(1.0 / np.sqrt(fan_in))
        out[name] = np.ascontiguousarray(a.astype(dtype))
    return out


_IGNORED = ("position_ids",)


def check_against_manifest(state: Dict[str, np.ndarray], manifest, what: str):
    missing = [k for k in manifest if k not in state]
    extra = [k for k in state if k not in manifest and not k.endswith(_IGNORED)]
    bad = [(k, tuple(state[k].shape), tuple(manifest[k])) for k in manifest
           if k in state and tuple(state[k].shape) != tuple(manifest[k])]
    if missing or extra or bad:
        raise ValueError(
            f"{what}: state dict does not match the architecture restated by this build "
            f"(missing {len(missing)}: {missing[:5]}; unexpected {len(extra)}: {extra[:5]}; "
            f"shape mismatches {len(bad)}: {bad[:5]})")


def load_safetensors(path: str) -> Dict[str, np.ndarray]:
    from safetensors.numpy import load_file
    return load_file(path)


def _first_existing(d, names):
    for n in names:
        p = os.path.join(d, n)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(f"none of {names} under {d}")


# Scheduler constants the engine's Karras / Euler tables are built from (csrc/engine.hip: karras_sigmas, k_euler_step; oracle/scheduler.py)
# = scheduler/scheduler_config.json of stable-video-diffusion-img2vid-xt.  A checkpoint whose file says otherwise must not load silently.
SCHEDULER_EXPECTED = {"_class_name": "EulerDiscreteScheduler", "sigma_min": 0.002, "sigma_max": 700.0, "prediction_type": "v_prediction",
                      "timestep_spacing": "leading", "timestep_type": "continuous", "use_karras_sigmas": True, "num_train_timesteps": 1000,
                      "steps_offset": 1, "interpolation_type": "linear"}
KARRAS_RHO = 7.0          # diffusers hard-codes rho = 7 in EulerDiscreteScheduler._convert_to_karras (not a config key)


def _read_json(path, what):
    if not os.path.exists(path):
        raise FileNotFoundError(f"{what}: {path} is missing (the reference's from_pretrained reads it, model/depthcrafter.py:18-29)")
    with open(path) as f:
        return json.load(f)


def _require(cfg, key, want, what):
    if key not in cfg:
        return                      # older config files omit keys whose default is the restated value
    got = cfg[key]
    same = (abs(float(got) - float(want)) <= 1e-9 * max(1.0, abs(float(want)))) if isinstance(want, float) else (got == want)
    if not same:
        raise ValueError(f"{what}: {key} = {got!r} but this build restates {want!r}; refusing to load a checkpoint the kernels were not written for")


def check_scheduler_config(cfg: dict, what="scheduler/scheduler_config.json"):
    for k, v in SCHEDULER_EXPECTED.items():
        _require(cfg, k, v, what)


def unet_cfg_from_config(cfg: dict) -> UNetCfg:
    """diffusers ``UNetSpatioTemporalConditionModel`` config.json -> UNetCfg.  Keys that select code this build does not have fail."""
    what = "UNet config.json"
    down = list(cfg.get("down_block_types", ["CrossAttnDownBlockSpatioTemporal"] * 3 + ["DownBlockSpatioTemporal"]))
    up = list(cfg.get("up_block_types", ["UpBlockSpatioTemporal"] + ["CrossAttnUpBlockSpatioTemporal"] * 3))
    ok_d = {"CrossAttnDownBlockSpatioTemporal": True, "DownBlockSpatioTemporal": False}
    ok_u = {"CrossAttnUpBlockSpatioTemporal": True, "UpBlockSpatioTemporal": False}
    if any(d not in ok_d for d in down) or any(u not in ok_u for u in up):
        raise ValueError(f"{what}: unsupported block types {down} / {up}")
    has = tuple(ok_d[d] for d in down)
    if tuple(ok_u[u] for u in up) != tuple(reversed(has)):
        raise ValueError(f"{what}: up_block_types {up} are not the mirror of down_block_types {down}")
    if has[-1] or not all(has[:-1]):
        raise ValueError(f"{what}: this build has attention on every level but the last ({down})")
    tl = cfg.get("transformer_layers_per_block", 1)
    if (tl if isinstance(tl, int) else max(tl)) != 1:
        raise ValueError(f"{what}: transformer_layers_per_block = {tl} (this build: 1)")
    heads = cfg.get("num_attention_heads", (5, 10, 20, 20))
    boc = tuple(cfg["block_out_channels"])
    heads = tuple(heads) if not isinstance(heads, int) else (heads,) * len(boc)
    lpb = cfg.get("layers_per_block", 2)
    if not isinstance(lpb, int):
        if len(set(lpb)) != 1:
            raise ValueError(f"{what}: per-level layers_per_block {lpb} not supported")
        lpb = lpb[0]
    if any(c % h or c // h != 64 for c, h in zip(boc, heads)):
        raise ValueError(f"{what}: attention head dim must be 64 (block_out_channels {boc}, num_attention_heads {heads})")
    d = UNetCfg()
    return UNetCfg(in_channels=cfg.get("in_channels", 8), out_channels=cfg.get("out_channels", 4), block_out_channels=boc, layers_per_block=lpb,
                   num_attention_heads=heads, cross_attention_dim=cfg.get("cross_attention_dim", 1024),
                   addition_time_embed_dim=cfg.get("addition_time_embed_dim", 256),
                   projection_class_embeddings_input_dim=cfg.get("projection_class_embeddings_input_dim", 768),
                   norm_groups=cfg.get("norm_num_groups", d.norm_groups),     # not a diffusers key (hard-coded 32 there); written by save_pretrained_layout for reduced configurations
                   down_has_attn=has)


def vae_cfg_from_config(cfg: dict) -> VAECfg:
    what = "VAE config.json"
    _require(cfg, "_class_name", "AutoencoderKLTemporalDecoder", what)
    if not cfg.get("force_upcast", True):
        raise ValueError(f"{what}: force_upcast = false - the engine runs the reference's float32 encoder; call ug_set_vae_encode_fp32(0) explicitly instead")
    return VAECfg(in_channels=cfg.get("in_channels", 3), out_channels=cfg.get("out_channels", 3), latent_channels=cfg.get("latent_channels", 4),
                  block_out_channels=tuple(cfg["block_out_channels"]), layers_per_block=cfg.get("layers_per_block", 2),
                  norm_groups=cfg.get("norm_num_groups", VAECfg().norm_groups), scaling_factor=float(cfg.get("scaling_factor", 0.18215)))


def clip_cfg_from_config(cfg: dict) -> CLIPCfg:
    what = "image_encoder config.json"
    _require(cfg, "hidden_act", "gelu", what)      # erf GELU (the engine's epilogue); "quick_gelu" towers would need another one
    return CLIPCfg(hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"], num_hidden_layers=cfg["num_hidden_layers"],
                   num_attention_heads=cfg["num_attention_heads"], image_size=cfg.get("image_size", 224), patch_size=cfg.get("patch_size", 14),
                   projection_dim=cfg.get("projection_dim", 1024), layer_norm_eps=float(cfg.get("layer_norm_eps", 1e-5)))


def load_pretrained(unet_path: str, pre_train_path: str, with_cfgs: bool = False):
    """What the reference's two ``from_pretrained`` calls read (model/depthcrafter.py:18-29), in the diffusers directory layout:

        unet_path/config.json, unet_path/diffusion_pytorch_model.safetensors                  (DepthCrafter UNet; fp32 file, cast to fp16)
        pre_train_path/vae/config.json, vae/diffusion_pytorch_model.fp16.safetensors           (variant="fp16")
        pre_train_path/image_encoder/config.json, image_encoder/model.fp16.safetensors
        pre_train_path/scheduler/scheduler_config.json

    -> (unet_state, vae_state, clip_state[, (UNetCfg, VAECfg, CLIPCfg)]).  The three config.json files give the architecture (so a reduced
    configuration loads too); every tensor is checked against the manifest of that architecture, and the scheduler file against the constants
    the engine's sigma tables are built from.  Any disagreement raises with a precise diff - nothing loads on a best-effort basis."""
    ucfg = unet_cfg_from_config(_read_json(os.path.join(unet_path, "config.json"), "UNet"))
    vcfg = vae_cfg_from_config(_read_json(os.path.join(pre_train_path, "vae", "config.json"), "VAE"))
    ccfg = clip_cfg_from_config(_read_json(os.path.join(pre_train_path, "image_encoder", "config.json"), "CLIP image encoder"))
    check_scheduler_config(_read_json(os.path.join(pre_train_path, "scheduler", "scheduler_config.json"), "scheduler"))
    if ucfg.cross_attention_dim != ccfg.projection_dim:
        raise ValueError(f"UNet cross_attention_dim {ucfg.cross_attention_dim} != image encoder projection_dim {ccfg.projection_dim}")
    u = load_safetensors(_first_existing(unet_path, [
        "diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors"]))
    v = load_safetensors(_first_existing(os.path.join(pre_train_path, "vae"), [
        "diffusion_pytorch_model.fp16.safetensors", "diffusion_pytorch_model.safetensors"]))
    c = load_safetensors(_first_existing(os.path.join(pre_train_path, "image_encoder"), [
        "model.fp16.safetensors", "model.safetensors"]))
    check_against_manifest(u, unet_manifest(ucfg), "UNet")
    check_against_manifest(v, vae_manifest(vcfg), "VAE")
    check_against_manifest(c, clip_manifest(ccfg), "CLIP image encoder")
    return (u, v, c, (ucfg, vcfg, ccfg)) if with_cfgs else (u, v, c)


def save_pretrained_layout(unet_path: str, pre_train_path: str, unet_state, vae_state, clip_state, cfgs=None, unet_dtype=np.float32):
    """Write states in the directory layout ``load_pretrained`` reads - the inverse, used by the tests (tiny configurations) and by
    tools that stage seeded weights as a checkpoint.  The UNet file is fp32 like the DepthCrafter release; VAE / image encoder are the
    ``variant="fp16"`` files.  config.json / scheduler_config.json carry exactly the keys the loader checks."""
    from safetensors.numpy import save_file
    u, v, c = cfgs or (UNetCfg(), VAECfg(), CLIPCfg())
    for d in (unet_path, os.path.join(pre_train_path, "vae"), os.path.join(pre_train_path, "image_encoder"), os.path.join(pre_train_path, "scheduler")):
        os.makedirs(d, exist_ok=True)
    save_file({k: np.ascontiguousarray(a.astype(unet_dtype)) for k, a in unet_state.items()}, os.path.join(unet_path, "diffusion_pytorch_model.safetensors"))
    save_file({k: np.ascontiguousarray(a.astype(np.float16)) for k, a in vae_state.items()},
              os.path.join(pre_train_path, "vae", "diffusion_pytorch_model.fp16.safetensors"))
    save_file({k: np.ascontiguousarray(a.astype(np.float16)) for k, a in clip_state.items()},
              os.path.join(pre_train_path, "image_encoder", "model.fp16.safetensors"))
    n = len(u.block_out_channels)
    down = ["CrossAttnDownBlockSpatioTemporal" if a else "DownBlockSpatioTemporal" for a in u.down_has_attn[:n]]
    up = ["CrossAttnUpBlockSpatioTemporal" if a else "UpBlockSpatioTemporal" for a in reversed(u.down_has_attn[:n])]
    with open(os.path.join(unet_path, "config.json"), "w") as f:
        json.dump({"_class_name": "UNetSpatioTemporalConditionModel", "in_channels": u.in_channels, "out_channels": u.out_channels,
                   "down_block_types": down, "up_block_types": up, "block_out_channels": list(u.block_out_channels),
                   "layers_per_block": u.layers_per_block, "num_attention_heads": list(u.num_attention_heads),
                   "cross_attention_dim": u.cross_attention_dim, "addition_time_embed_dim": u.addition_time_embed_dim,
                   "projection_class_embeddings_input_dim": u.projection_class_embeddings_input_dim, "transformer_layers_per_block": 1,
                   "norm_num_groups": u.norm_groups, "num_frames": 25, "sample_size": 96}, f, indent=1)
    with open(os.path.join(pre_train_path, "vae", "config.json"), "w") as f:
        json.dump({"_class_name": "AutoencoderKLTemporalDecoder", "in_channels": v.in_channels, "out_channels": v.out_channels,
                   "latent_channels": v.latent_channels, "block_out_channels": list(v.block_out_channels), "layers_per_block": v.layers_per_block,
                   "norm_num_groups": v.norm_groups, "scaling_factor": v.scaling_factor, "force_upcast": True, "sample_size": 768}, f, indent=1)
    with open(os.path.join(pre_train_path, "image_encoder", "config.json"), "w") as f:
        json.dump({"architectures": ["CLIPVisionModelWithProjection"], "hidden_act": "gelu", "hidden_size": c.hidden_size,
                   "intermediate_size": c.intermediate_size, "num_hidden_layers": c.num_hidden_layers, "num_attention_heads": c.num_attention_heads,
                   "image_size": c.image_size, "patch_size": c.patch_size, "projection_dim": c.projection_dim, "layer_norm_eps": c.layer_norm_eps}, f, indent=1)
    with open(os.path.join(pre_train_path, "scheduler", "scheduler_config.json"), "w") as f:
        json.dump({**SCHEDULER_EXPECTED, "beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear"}, f, indent=1)


# ------------------------------------------------------------------------------------------------------------------
# StableNormal (BASELINE configs[3]; reference model/stablenormal.py:16 loads it through torch.hub - un-vendored).
# Restated components (oracle/stablenormal.py, DESIGN.md "StableNormal uncertainty register"): SD-2.1-class UNet2DConditionModel
# x 2 (YOSO one-step + refinement), ControlNet trunk x 2 (image-latent + DINO-guided), SD AutoencoderKL, DINOv2 ViT-L/14.
# ------------------------------------------------------------------------------------------------------------------
@dataclass
class SDUNetCfg:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    num_attention_heads: Tuple[int, ...] = (5, 10, 20, 20)
    cross_attention_dim: int = 1024
    norm_groups: int = 32
    down_has_attn: Tuple[bool, ...] = (True, True, True, False)


@dataclass
class DinoCfg:
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    image_size: int = 224
    patch_size: int = 14
    layer_norm_eps: float = 1e-6


def tiny_sn_cfgs():
    """Small StableNormal configuration with the full topology (every block type of both UNets, both ControlNets, the 2-D VAE
    decoder and the DINO tower) for parity tests the CPU oracle finishes in seconds."""
    u = SDUNetCfg(block_out_channels=(64, 128, 128, 128), num_attention_heads=(1, 2, 2, 2), cross_attention_dim=48, norm_groups=16)
    v = VAECfg(block_out_channels=(32, 64, 64, 64), norm_groups=8)
    d = DinoCfg(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1)
    return u, v, d


def _sd_transformer(m, p, ch, cross):
    m.norm(p + ".norm", ch); m.lin(p + ".proj_in", ch, ch)
    b = p + ".transformer_blocks.0"
    m.norm(b + ".norm1", ch); _attn(m, b + ".attn1", ch, ch)
    m.norm(b + ".norm2", ch); _attn(m, b + ".attn2", ch, ch, cross)
    m.norm(b + ".norm3", ch); _ff(m, b + ".ff", ch)
    m.lin(p + ".proj_out", ch, ch)


def _sd_res(m, p, cin, cout, temb):
    m.norm(p + ".norm1", cin); m.conv2(p + ".conv1", cin, cout)
    m.lin(p + ".time_emb_proj", temb, cout)
    m.norm(p + ".norm2", cout); m.conv2(p + ".conv2", cout, cout)
    if cin != cout:
        m.conv2(p + ".conv_shortcut", cin, cout, k=1)


def _sd_trunk(m, cfg):
    boc = cfg.block_out_channels
    n, temb, x = len(boc), boc[0] * 4, cfg.cross_attention_dim
    m.conv2("conv_in", cfg.in_channels, boc[0])
    m.lin("time_embedding.linear_1", boc[0], temb); m.lin("time_embedding.linear_2", temb, temb)
    ch = boc[0]
    for i in range(n):
        for j in range(cfg.layers_per_block):
            _sd_res(m, f"down_blocks.{i}.resnets.{j}", ch if j == 0 else boc[i], boc[i], temb)
            if cfg.down_has_attn[i]:
                _sd_transformer(m, f"down_blocks.{i}.attentions.{j}", boc[i], x)
        if i != n - 1:
            m.conv2(f"down_blocks.{i}.downsamplers.0.conv", boc[i], boc[i])
        ch = boc[i]
    _sd_res(m, "mid_block.resnets.0", boc[-1], boc[-1], temb)
    _sd_transformer(m, "mid_block.attentions.0", boc[-1], x)
    _sd_res(m, "mid_block.resnets.1", boc[-1], boc[-1], temb)


def sd_unet_manifest(cfg: SDUNetCfg = SDUNetCfg()) -> "OrderedDict[str, tuple]":
    """diffusers UNet2DConditionModel (SD 2.1): 865 910 724 parameters at the default configuration."""
    m = _M()
    _sd_trunk(m, cfg)
    boc = cfg.block_out_channels
    n, temb, x = len(boc), boc[0] * 4, cfg.cross_attention_dim
    rev, rattn = list(reversed(boc)), list(reversed(cfg.down_has_attn))
    out, L = rev[0], cfg.layers_per_block + 1
    for i in range(n):
        prev, out = out, rev[i]
        cin = rev[min(i + 1, n - 1)]
        for j in range(L):
            _sd_res(m, f"up_blocks.{i}.resnets.{j}", (prev if j == 0 else out) + (cin if j == L - 1 else out), out, temb)
            if rattn[i]:
                _sd_transformer(m, f"up_blocks.{i}.attentions.{j}", out, x)
        if i != n - 1:
            m.conv2(f"up_blocks.{i}.upsamplers.0.conv", out, out)
    m.norm("conv_norm_out", boc[0]); m.conv2("conv_out", boc[0], cfg.out_channels)
    return m


def controlnet_manifest(cfg: SDUNetCfg = SDUNetCfg(), dino_dim: int = 0) -> "OrderedDict[str, tuple]":
    """diffusers ControlNetModel trunk + zero convs; the image latent is its `sample` (no pixel-space conditioning stem); the DINO
    variant carries `dino_controlnet_cond_embedding` (Linear dino_dim -> conv_in width)."""
    m = _M()
    _sd_trunk(m, cfg)
    boc = cfg.block_out_channels
    chans = [boc[0]]
    for i, c in enumerate(boc):
        chans += [c] * cfg.layers_per_block + ([c] if i != len(boc) - 1 else [])
    for k, c in enumerate(chans):
        m.conv2(f"controlnet_down_blocks.{k}", c, c, k=1)
    m.conv2("controlnet_mid_block", boc[-1], boc[-1], k=1)
    if dino_dim:
        m.lin("dino_controlnet_cond_embedding", dino_dim, boc[0])
    return m


def sd_vae_manifest(cfg: VAECfg = VAECfg()) -> "OrderedDict[str, tuple]":
    """diffusers AutoencoderKL (SD): the encoder of vae_manifest + the plain 2-D decoder + quant / post_quant convs (83 653 863 parameters)."""
    m = _M()
    full = vae_manifest(cfg)
    for k, v in full.items():
        if k.startswith("encoder.") or k.startswith("quant_conv"):
            m[k] = v
    boc, L = cfg.block_out_channels, cfg.layers_per_block
    m.conv2("post_quant_conv", cfg.latent_channels, cfg.latent_channels, k=1)
    m.conv2("decoder.conv_in", cfg.latent_channels, boc[-1])
    _res2d(m, "decoder.mid_block.resnets.0", boc[-1], boc[-1])
    _vae_attn(m, "decoder.mid_block.attentions.0", boc[-1])
    _res2d(m, "decoder.mid_block.resnets.1", boc[-1], boc[-1])
    rev, out = list(reversed(boc)), boc[-1]
    for i in range(len(boc)):
        prev, out = out, rev[i]
        for j in range(L + 1):
            _res2d(m, f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else out, out)
        if i != len(boc) - 1:
            m.conv2(f"decoder.up_blocks.{i}.upsamplers.0.conv", out, out)
    m.norm("decoder.conv_norm_out", boc[0]); m.conv2("decoder.conv_out", boc[0], cfg.out_channels)
    return m


def dino_manifest(cfg: DinoCfg = DinoCfg()) -> "OrderedDict[str, tuple]":
    """dinov2 hub naming (ViT-L/14 + LayerScale); pos_embed is held at the tower's own grid (image_size / patch_size)^2 + 1 -
    a checkpoint's 37x37 table is resampled by load_stablenormal_pretrained."""
    m = _M()
    d, g = cfg.hidden_size, cfg.image_size // cfg.patch_size
    m["cls_token"] = (1, 1, d)
    m["pos_embed"] = (1, 1 + g * g, d)
    m["patch_embed.proj.weight"] = (d, 3, cfg.patch_size, cfg.patch_size); m["patch_embed.proj.bias"] = (d,)
    for i in range(cfg.num_hidden_layers):
        p = f"blocks.{i}"
        m.norm(p + ".norm1", d); m.lin(p + ".attn.qkv", d, 3 * d); m.lin(p + ".attn.proj", d, d); m[p + ".ls1.gamma"] = (d,)
        m.norm(p + ".norm2", d); m.lin(p + ".mlp.fc1", d, cfg.intermediate_size); m.lin(p + ".mlp.fc2", cfg.intermediate_size, d)
        m[p + ".ls2.gamma"] = (d,)
    m.norm("norm", d)
    return m


def dinov2_hf_to_hub(state):
    """A `transformers.Dinov2Model` state dict -> dinov2 hub naming (what dino_manifest lists): the fused `attn.qkv` is query | key | value stacked in that order.
    Pinned against transformers by tests/test_oracle_structure.py::test_dinov2_oracle_matches_transformers."""
    out = OrderedDict()
    cat = lambda parts: np.concatenate([np.asarray(x) for x in parts], 0)
    layers = sorted({int(k.split(".")[2]) for k in state if k.startswith("encoder.layer.")})
    out["cls_token"] = state["embeddings.cls_token"]; out["pos_embed"] = state["embeddings.position_embeddings"]
    out["patch_embed.proj.weight"] = state["embeddings.patch_embeddings.projection.weight"]
    out["patch_embed.proj.bias"] = state["embeddings.patch_embeddings.projection.bias"]
    for i in layers:
        h, b = f"encoder.layer.{i}", f"blocks.{i}"
        for wb in ("weight", "bias"):
            out[f"{b}.norm1.{wb}"] = state[f"{h}.norm1.{wb}"]; out[f"{b}.norm2.{wb}"] = state[f"{h}.norm2.{wb}"]
            out[f"{b}.attn.qkv.{wb}"] = cat([state[f"{h}.attention.attention.{n}.{wb}"] for n in ("query", "key", "value")])
            out[f"{b}.attn.proj.{wb}"] = state[f"{h}.attention.output.dense.{wb}"]
            out[f"{b}.mlp.fc1.{wb}"] = state[f"{h}.mlp.fc1.{wb}"]; out[f"{b}.mlp.fc2.{wb}"] = state[f"{h}.mlp.fc2.{wb}"]
        out[f"{b}.ls1.gamma"] = state[f"{h}.layer_scale1.lambda1"]; out[f"{b}.ls2.gamma"] = state[f"{h}.layer_scale2.lambda1"]
    out["norm.weight"] = state["layernorm.weight"]; out["norm.bias"] = state["layernorm.bias"]
    return out


def resample_dino_pos_embed(pe, g_dst: int):
    """DINOv2 position table [1, 1 + g_src^2, D] -> [1, 1 + g_dst^2, D]: the class row kept, the patch grid resampled bicubically (align_corners False,
    target SIZE given) - transformers' `Dinov2Embeddings.interpolate_pos_encoding` (pinned by test_dinov2_position_table_resampling_matches_transformers)."""
    pe = np.asarray(pe, np.float32)
    g_src = int(round((pe.shape[1] - 1) ** 0.5))
    if g_src == g_dst:
        return pe
    import torch
    grid = torch.from_numpy(pe[0, 1:]).reshape(1, g_src, g_src, -1).permute(0, 3, 1, 2)
    grid = torch.nn.functional.interpolate(grid, size=(g_dst, g_dst), mode="bicubic", align_corners=False)
    return np.concatenate([pe[:, :1], grid.permute(0, 2, 3, 1).reshape(1, g_dst * g_dst, -1).numpy()], 1)


def load_stablenormal_pretrained(model_dir: str, cfgs=None):
    """Checkpoint directory -> ({component: state}, prompt_embeds [77,1024]).  Expected layout (diffusers-style, one sub-directory per
    component; S11): ``vae/``, ``unet_yoso/``, ``controlnet_yoso/``, ``unet/``, ``controlnet_dino/`` each with
    ``diffusion_pytorch_model[.fp16].safetensors``; ``dino/model.safetensors`` (dinov2 hub naming); ``text_encoder/`` + ``tokenizer/``
    (transformers CLIPTextModel - run ONCE here, on the host, for the fixed prompt) or a precomputed ``prompt_embeds.npy``.
    Every state dict is checked against the manifest; DINO's 37x37 position table is resampled (bicubic) to the tower's grid."""
    cfgs = cfgs or (SDUNetCfg(), VAECfg(), DinoCfg())
    names = ["diffusion_pytorch_model.fp16.safetensors", "diffusion_pytorch_model.safetensors", "model.fp16.safetensors", "model.safetensors"]
    man = {"vae": sd_vae_manifest(cfgs[1]), "unet_yoso": sd_unet_manifest(cfgs[0]), "controlnet_yoso": controlnet_manifest(cfgs[0]),
           "unet": sd_unet_manifest(cfgs[0]), "controlnet_dino": controlnet_manifest(cfgs[0], cfgs[2].hidden_size), "dino": dino_manifest(cfgs[2])}
    states = {}
    for comp, m in man.items():
        st = load_safetensors(_first_existing(os.path.join(model_dir, comp), names))
        if comp == "dino":
            if "embeddings.cls_token" in st:                      # a transformers Dinov2Model checkpoint instead of the hub's naming
                st = dinov2_hf_to_hub(st)
            st = {k: v for k, v in st.items() if k != "mask_token"}
            st["pos_embed"] = resample_dino_pos_embed(st["pos_embed"], cfgs[2].image_size // cfgs[2].patch_size)
        check_against_manifest(st, m, f"StableNormal {comp}")
        states[comp] = st
    pe_file = os.path.join(model_dir, "prompt_embeds.npy")
    if os.path.exists(pe_file):
        prompt = np.load(pe_file).astype(np.float32).reshape(77, -1)
    else:
        import torch
        from transformers import CLIPTextModel, CLIPTokenizer
        tok = CLIPTokenizer.from_pretrained(os.path.join(model_dir, "tokenizer"))
        enc = CLIPTextModel.from_pretrained(os.path.join(model_dir, "text_encoder")).eval()
        ids = tok("The normal map", padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        with torch.no_grad():
            prompt = enc(ids)[0][0].float().numpy()
    return states, prompt
