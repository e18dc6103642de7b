"""ctypes binding of libunigeo_hip.so (C ABI: include/unigeo_hip.h).

There is deliberately NO fallback: if the shared library is missing or no MI355X is visible
the import / Engine construction raises.  numpy arrays in, numpy arrays out; torch never
crosses this boundary.
"""
import ctypes as C
import dataclasses
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UG_LIB_PATH") or os.path.join(_HERE, "csrc", "libunigeo_hip.so")   # UG_LIB_PATH: A/B a second build of the SAME library (tools/ab)

EXPORTS = [
    "ug_unet_config_default", "ug_vae_config_default", "ug_clip_config_default",
    "ug_create", "ug_destroy", "ug_last_error", "ug_workspace_peak",
    "ug_load_tensor", "ug_bind_unet", "ug_bind_vae", "ug_bind_clip",
    "ug_dc_set_inputs", "ug_dc_run", "ug_dc_run_windows", "ug_dc_get_outputs", "ug_dc_device_ptrs", "ug_dc_set_trace", "ug_set_vae_encode_fp32", "ug_set_concurrency", "ug_set_coscheduled", "ug_set_fp8_linears", "ug_op_linear_mx8", "ug_set_ff_fused", "ug_op_ff", "ug_op_ln_ff", "ug_bench_ff", "ug_bench_flash", "ug_tune_flash", "ug_tune_ff",
    "ug_eval_depth", "ug_eval_normal", "ug_clip_embed", "ug_vae_encode", "ug_vae_decode", "ug_unet_forward", "ug_normals_from_depth",
    "ug_op_linear", "ug_op_conv", "ug_op_conv_gn", "ug_op_groupnorm", "ug_op_layernorm", "ug_op_flash_attn",
    "ug_op_temporal_attn", "ug_op_attention_generic", "ug_op_flash_attn_dh", "ug_op_euler_step",
    "ug_bind_stablenormal", "ug_sn_run", "ug_sn_unet_forward", "ug_sn_dino", "ug_sn_vae_decode", "ug_sn_vae_encode", "ug_resize_bilinear",
    "ug_profile_begin", "ug_profile_begin_shapes", "ug_profile_end", "ug_bench_gemm", "ug_bench_groupnorm", "ug_bench_mfma_peak", "ug_tune_force",
]


class UNetConfigC(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("num_levels", C.c_int),
                ("block_out_channels", C.c_int * 8), ("num_attention_heads", C.c_int * 8),
                ("down_has_attn", C.c_int * 8),
                ("layers_per_block", C.c_int), ("cross_attention_dim", C.c_int),
                ("addition_time_embed_dim", C.c_int), ("projection_class_embeddings_input_dim", C.c_int),
                ("norm_groups", C.c_int),
                ("eps_cross_attn_blocks", C.c_float), ("eps_plain_down_block", C.c_float),
                ("eps_mid_block", C.c_float), ("eps_up_blocks", C.c_float)]


class VAEConfigC(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("latent_channels", C.c_int),
                ("num_levels", C.c_int), ("block_out_channels", C.c_int * 8),
                ("layers_per_block", C.c_int), ("norm_groups", C.c_int), ("scaling_factor", C.c_float)]


class CLIPConfigC(C.Structure):
    _fields_ = [("hidden_size", C.c_int), ("intermediate_size", C.c_int), ("num_hidden_layers", C.c_int),
                ("num_attention_heads", C.c_int), ("image_size", C.c_int), ("patch_size", C.c_int),
                ("projection_dim", C.c_int), ("layer_norm_eps", C.c_float)]


_lib = None


def load_library():
    """dlopen the in-tree library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C unigeo_amd/csrc`).  The MI355X path has no CPU / PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    fp, ip, vp = C.POINTER(C.c_float), C.c_int, C.c_void_p
    lib.ug_create.restype = vp
    lib.ug_create.argtypes = [ip, C.c_size_t, C.c_size_t]
    lib.ug_destroy.argtypes = [vp]
    lib.ug_last_error.restype = C.c_char_p
    lib.ug_last_error.argtypes = [vp]
    lib.ug_workspace_peak.restype = C.c_size_t
    lib.ug_workspace_peak.argtypes = [vp]
    lib.ug_load_tensor.argtypes = [vp, C.c_char_p, ip, ip, C.POINTER(C.c_int64), vp]
    lib.ug_bind_unet.argtypes = [vp, C.POINTER(UNetConfigC)]
    lib.ug_bind_vae.argtypes = [vp, C.POINTER(VAEConfigC)]
    lib.ug_bind_clip.argtypes = [vp, C.POINTER(CLIPConfigC)]
    lib.ug_dc_set_inputs.argtypes = [vp, vp, ip, ip, ip, vp, vp, vp]
    lib.ug_dc_run.argtypes = [vp, ip, ip, ip]
    lib.ug_dc_run_windows.argtypes = [vp, ip, ip, ip, ip, ip]
    lib.ug_dc_get_outputs.argtypes = [vp, vp, vp, vp]
    lib.ug_dc_set_trace.argtypes = [vp, vp, ip]
    lib.ug_set_vae_encode_fp32.argtypes = [vp, ip]
    try:
        lib.ug_set_fp8_linears.argtypes = [vp, ip]
        lib.ug_set_concurrency.argtypes = [vp, ip]
        lib.ug_set_coscheduled.argtypes = [vp, ip]
        lib.ug_set_ff_fused.argtypes = [vp, ip]
        lib.ug_bench_ff.argtypes = [vp, ip, ip, ip, ip, vp]
        lib.ug_bench_flash.argtypes = [vp, ip, ip, ip, ip, ip, vp]
        lib.ug_tune_flash.argtypes = [vp, ip]
        lib.ug_tune_ff.argtypes = [vp, ip]
        lib.ug_op_ff.argtypes = [vp, vp, ip, ip, vp, vp, vp, vp, vp, C.c_float, C.c_float, ip, vp]
        lib.ug_op_ln_ff.argtypes = [vp, vp, ip, ip, vp, vp, C.c_float, vp, ip, vp, vp, vp, vp, C.c_float, C.c_float, ip, vp]
        lib.ug_op_linear_mx8.argtypes = [vp, vp, ip, ip, vp, ip, vp, ip, vp, vp, vp]
    except AttributeError:
        if not os.environ.get("UG_LIB_PATH"):      # only an explicitly selected OLDER build (tools/ab A/B runs) may lack these
            raise
    lib.ug_dc_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.ug_eval_depth.argtypes = [vp, vp, vp, vp, C.c_long, C.c_float, vp]
    lib.ug_eval_normal.argtypes = [vp, vp, vp, vp, C.c_long, vp]
    lib.ug_clip_embed.argtypes = [vp, vp, ip, ip, ip, vp]
    lib.ug_vae_encode.argtypes = [vp, vp, ip, ip, ip, vp]
    lib.ug_vae_decode.argtypes = [vp, vp, ip, ip, ip, vp]
    lib.ug_unet_forward.argtypes = [vp, vp, ip, ip, ip, C.c_float, vp, vp]
    lib.ug_normals_from_depth.argtypes = [vp, vp, vp, ip, ip, ip, vp]
    lib.ug_op_linear.argtypes = [vp, vp, ip, ip, vp, ip, vp, vp, C.c_float, C.c_float, ip, ip, vp]
    lib.ug_op_conv.argtypes = [vp, vp, ip, vp, ip, ip, ip, ip, vp, vp, ip, ip, ip, ip, ip, ip, ip, vp]
    lib.ug_op_groupnorm.argtypes = [vp, vp, ip, vp, ip, ip, ip, ip, C.c_float, ip, ip, vp, vp, vp]
    lib.ug_op_conv_gn.argtypes = [vp, vp, ip, ip, ip, ip, vp, vp, vp, ip, ip, ip, ip, C.c_float, ip, vp, vp, vp, vp, vp, vp]
    lib.ug_op_layernorm.argtypes = [vp, vp, ip, ip, C.c_float, vp, vp, vp, ip, vp, vp]
    lib.ug_op_flash_attn.argtypes = [vp, vp, ip, ip, ip, vp]
    lib.ug_op_temporal_attn.argtypes = [vp, vp, ip, ip, ip, vp]
    lib.ug_op_attention_generic.argtypes = [vp, vp, ip, ip, ip, ip, vp]
    lib.ug_op_flash_attn_dh.argtypes = [vp, vp, ip, ip, ip, ip, vp]
    lib.ug_op_euler_step.argtypes = [vp, vp, vp, C.c_long, C.c_float, C.c_float]
    lib.ug_bind_stablenormal.argtypes = [vp, C.POINTER(UNetConfigC), C.POINTER(VAEConfigC), C.POINTER(CLIPConfigC)]
    lib.ug_sn_run.argtypes = [vp, vp, ip, ip, ip, vp, C.c_float, ip, vp, vp, vp, vp]
    lib.ug_sn_unet_forward.argtypes = [vp, ip, vp, vp, ip, ip, ip, C.c_float, C.c_float, vp, vp, ip, vp]
    lib.ug_sn_dino.argtypes = [vp, vp, ip, ip, ip, vp]
    lib.ug_sn_vae_decode.argtypes = [vp, vp, ip, ip, ip, vp]
    lib.ug_sn_vae_encode.argtypes = [vp, vp, ip, ip, ip, vp]
    lib.ug_resize_bilinear.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, ip, vp]
    lib.ug_profile_begin.argtypes = [vp]
    lib.ug_bench_gemm.argtypes = [vp] + [ip] * 16 + [vp]
    lib.ug_bench_groupnorm.argtypes = [vp, ip, ip, ip, ip, ip, ip, ip, vp]
    lib.ug_tune_force.argtypes = [vp, ip, ip]
    lib.ug_profile_begin_shapes.argtypes = [vp]
    lib.ug_profile_end.restype = C.c_char_p
    lib.ug_profile_end.argtypes = [vp]
    _lib = lib
    return lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _fill8(dst, vals):
    for i in range(8):
        dst[i] = int(vals[i]) if i < len(vals) else 0


class Engine:
    """One context = one GPU.  Mirrors, at the C-ABI level, what the reference's pipeline object
    offers at /root/reference/model/depthcrafter.py:24-34,80-90."""

    def __init__(self, device_id=0, workspace_bytes=2 << 30, persist_bytes=1 << 30):
        self.lib = load_library()
        self.ctx = self.lib.ug_create(int(device_id), int(workspace_bytes), int(persist_bytes))
        if not self.ctx:
            raise RuntimeError("ug_create failed: " + self.lib.ug_last_error(None).decode())
        self.device_id = device_id

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.ug_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError("libunigeo_hip: " + self.lib.ug_last_error(self.ctx).decode())

    # ---- weights
    def load_state(self, prefix, state):
        for name, arr in state.items():
            a = np.ascontiguousarray(arr)
            if a.dtype == np.float16:
                dt = 0
            elif a.dtype == np.float32:
                dt = 1
            else:
                a = a.astype(np.float32); dt = 1
            shape = (C.c_int64 * max(a.ndim, 1))(*(a.shape if a.ndim else (1,)))
            self._ck(self.lib.ug_load_tensor(self.ctx, (prefix + name).encode(), dt, max(a.ndim, 1), shape, _ptr(a)))

    def bind_unet(self, cfg):
        c = UNetConfigC()
        c.in_channels, c.out_channels = cfg.in_channels, cfg.out_channels
        c.num_levels = len(cfg.block_out_channels)
        _fill8(c.block_out_channels, cfg.block_out_channels)
        _fill8(c.num_attention_heads, cfg.num_attention_heads)
        _fill8(c.down_has_attn, [int(b) for b in cfg.down_has_attn])
        c.layers_per_block, c.cross_attention_dim = cfg.layers_per_block, cfg.cross_attention_dim
        c.addition_time_embed_dim = cfg.addition_time_embed_dim
        c.projection_class_embeddings_input_dim = cfg.projection_class_embeddings_input_dim
        c.norm_groups = cfg.norm_groups
        c.eps_cross_attn_blocks, c.eps_plain_down_block = cfg.eps_cross_attn_blocks, cfg.eps_plain_down_block
        c.eps_mid_block, c.eps_up_blocks = cfg.eps_mid_block, cfg.eps_up_blocks
        self._ck(self.lib.ug_bind_unet(self.ctx, C.byref(c)))
        self.unet_cfg = cfg

    def bind_vae(self, cfg):
        c = VAEConfigC()
        c.in_channels, c.out_channels, c.latent_channels = cfg.in_channels, cfg.out_channels, cfg.latent_channels
        c.num_levels = len(cfg.block_out_channels)
        _fill8(c.block_out_channels, cfg.block_out_channels)
        c.layers_per_block, c.norm_groups, c.scaling_factor = cfg.layers_per_block, cfg.norm_groups, cfg.scaling_factor
        self._ck(self.lib.ug_bind_vae(self.ctx, C.byref(c)))
        self.vae_cfg = cfg

    def bind_clip(self, cfg):
        c = CLIPConfigC()
        c.hidden_size, c.intermediate_size = cfg.hidden_size, cfg.intermediate_size
        c.num_hidden_layers, c.num_attention_heads = cfg.num_hidden_layers, cfg.num_attention_heads
        c.image_size, c.patch_size, c.projection_dim = cfg.image_size, cfg.patch_size, cfg.projection_dim
        c.layer_norm_eps = cfg.layer_norm_eps
        self._ck(self.lib.ug_bind_clip(self.ctx, C.byref(c)))
        self.clip_cfg = cfg

    # ---- StableNormal
    def bind_stablenormal(self, ucfg, vcfg, dcfg):
        u = UNetConfigC()
        u.in_channels, u.out_channels, u.num_levels = ucfg.in_channels, ucfg.out_channels, len(ucfg.block_out_channels)
        _fill8(u.block_out_channels, ucfg.block_out_channels); _fill8(u.num_attention_heads, ucfg.num_attention_heads)
        _fill8(u.down_has_attn, [int(b) for b in ucfg.down_has_attn])
        u.layers_per_block, u.cross_attention_dim, u.norm_groups = ucfg.layers_per_block, ucfg.cross_attention_dim, ucfg.norm_groups
        v = VAEConfigC()
        v.in_channels, v.out_channels, v.latent_channels = vcfg.in_channels, vcfg.out_channels, vcfg.latent_channels
        v.num_levels = len(vcfg.block_out_channels)
        _fill8(v.block_out_channels, vcfg.block_out_channels)
        v.layers_per_block, v.norm_groups, v.scaling_factor = vcfg.layers_per_block, vcfg.norm_groups, vcfg.scaling_factor
        d = CLIPConfigC()
        d.hidden_size, d.intermediate_size, d.num_hidden_layers = dcfg.hidden_size, dcfg.intermediate_size, dcfg.num_hidden_layers
        d.num_attention_heads, d.image_size, d.patch_size, d.layer_norm_eps = dcfg.num_attention_heads, dcfg.image_size, dcfg.patch_size, dcfg.layer_norm_eps
        self._ck(self.lib.ug_bind_stablenormal(self.ctx, C.byref(u), C.byref(v), C.byref(d)))
        self.sn_cfgs = (ucfg, vcfg, dcfg)

    def sn_run(self, images, prompt_embeds, yoso_t, timesteps, ca, cb):
        f = _f32(images); B, H, W, _ = f.shape
        pe = _f32(prompt_embeds).reshape(77, self.sn_cfgs[0].cross_attention_dim)
        ts, a, b = _f32(timesteps), _f32(ca), _f32(cb)
        out = np.empty((B, H, W, 3), np.float32)
        self._ck(self.lib.ug_sn_run(self.ctx, _ptr(f), B, H, W, _ptr(pe), float(yoso_t), int(ts.size), _ptr(ts), _ptr(a), _ptr(b), _ptr(out)))
        return out

    def sn_unet_forward(self, which, sample, t_unet, prompt_embeds, zimg=None, t_ctrl=0.0, dino_tokens=None, use_ctrl=False):
        s = _f32(sample); B, _, h, w = s.shape
        z = None if zimg is None else _f32(zimg)
        d = None if dino_tokens is None else _f32(dino_tokens)
        pe = _f32(prompt_embeds)
        out = np.empty((B, 4, h, w), np.float32)
        self._ck(self.lib.ug_sn_unet_forward(self.ctx, int(which), _ptr(s), _ptr(z), B, h, w, float(t_unet), float(t_ctrl), _ptr(pe), _ptr(d),
                                             int(bool(use_ctrl)), _ptr(out)))
        return out

    def sn_dino(self, images):
        f = _f32(images); B, H, W, _ = f.shape
        dc = self.sn_cfgs[2]; g = dc.image_size // dc.patch_size
        out = np.empty((B, g * g, dc.hidden_size), np.float32)
        self._ck(self.lib.ug_sn_dino(self.ctx, _ptr(f), B, H, W, _ptr(out)))
        return out

    def sn_vae_decode(self, z):
        z = _f32(z); B, _, h, w = z.shape
        out = np.empty((B, 8 * h, 8 * w, 3), np.float32)
        self._ck(self.lib.ug_sn_vae_decode(self.ctx, _ptr(z), B, h, w, _ptr(out)))
        return out

    def resize_bilinear(self, x_bhwc, Ho, Wo, normalise=False):
        """torch F.interpolate(mode="bilinear", antialias=True, align_corners=False) on the device; [B,Hi,Wi,C<=4] f32 -> [B,Ho,Wo,C]."""
        x = _f32(x_bhwc); B, Hi, Wi, Cc = x.shape
        out = np.empty((B, int(Ho), int(Wo), Cc), np.float32)
        self._ck(self.lib.ug_resize_bilinear(self.ctx, _ptr(x), B, Hi, Wi, Cc, int(Ho), int(Wo), int(bool(normalise)), _ptr(out)))
        return out

    def sn_vae_encode(self, img_m11):
        f = _f32(img_m11); B, H, W, _ = f.shape
        out = np.empty((B, 4, H // 8, W // 8), np.float32)
        self._ck(self.lib.ug_sn_vae_encode(self.ctx, _ptr(f), B, H, W, _ptr(out)))
        return out

    # ---- pipeline
    def set_inputs(self, frames, noise_latents, noise_aug, intrinsics=None):
        f = _f32(frames)
        T, H, W, _ = f.shape
        nl, na = _f32(noise_latents).reshape(T, 4, H // 8, W // 8), _f32(noise_aug).reshape(T, 3, H, W)
        k = None if intrinsics is None else _f32(intrinsics).reshape(T, 3, 3)
        self._ck(self.lib.ug_dc_set_inputs(self.ctx, _ptr(f), T, H, W, _ptr(nl), _ptr(na), _ptr(k)))
        self._shape = (T, H, W)

    def run(self, steps, decode_chunk=8, with_normals=False, window=0, overlap=0):
        if window:
            self._ck(self.lib.ug_dc_run_windows(self.ctx, int(steps), int(decode_chunk), int(bool(with_normals)), int(window), int(overlap)))
        else:
            self._ck(self.lib.ug_dc_run(self.ctx, int(steps), int(decode_chunk), int(bool(with_normals))))

    def set_coscheduled(self, on=True):
        """This context shares the GPU with another clip in flight (a second context): drop the heuristics that fill the last round of one kernel at the
        price of extra launches / work (fused feed-forward tail split, last-round fill factor of the tile planner)."""
        self._ck(self.lib.ug_set_coscheduled(self.ctx, int(bool(on))))

    def set_concurrency(self, lanes=2):
        """Independent chunks (VAE encode / decode chunks, CLIP tower) in flight on separate HIP streams; 1 = serial.  Bit-identical outputs."""
        self._ck(self.lib.ug_set_concurrency(self.ctx, int(lanes)))

    def set_vae_encode_fp32(self, on=True):
        """True (default) = the reference's float32 VAE encoder (force_upcast); False = fp16 storage like the decoder."""
        self._ck(self.lib.ug_set_vae_encode_fp32(self.ctx, int(bool(on))))

    def bench_flash(self, B, H, S, variant=0, iters=20):
        out = np.zeros(1, np.float32)
        self._ck(self.lib.ug_bench_flash(self.ctx, B, H, S, int(variant), int(iters), _ptr(out)))
        return float(out[0])

    def bench_ff(self, M, C, fused=True, iters=20):
        out = np.zeros(1, np.float32)
        self._ck(self.lib.ug_bench_ff(self.ctx, int(M), int(C), int(bool(fused)), int(iters), _ptr(out)))
        return float(out[0])

    def set_ff_fused(self, on=True, prenorm=True):
        """Fused GEGLU feed-forward kernel of the narrow transformer blocks (on) and its LayerNorm inside the kernel (prenorm); A/B aid."""
        self._ck(self.lib.ug_set_ff_fused(self.ctx, (1 if on else 0) | (2 if on and prenorm else 0)))

    def tune_ff(self, variant=0):
        """Per-context A/B aid: fused feed-forward kernel form, 0 = cross-tile prefetch (default), 1 = without it."""
        self._ck(self.lib.ug_tune_ff(self.ctx, int(variant)))

    def tune_flash(self, variant=-1):
        """Per-context test aid: flash-attention variant mask of this context's launches (-1 = default 23)."""
        self._ck(self.lib.ug_tune_flash(self.ctx, int(variant)))

    def op_ln_ff(self, X, gamma, beta, W1, b1, W2, b2, addvec=None, rows_per_vec=1, eps=1e-5, c0=1.0, c1=1.0, mode=2):
        X = _f32(X); M, Cc = X.shape
        av = None if addvec is None else _f32(addvec)
        out = np.empty((M, Cc), np.float32)
        self._ck(self.lib.ug_op_ln_ff(self.ctx, _ptr(X), M, Cc, _ptr(_f32(gamma)), _ptr(_f32(beta)), float(eps), _ptr(av), int(rows_per_vec),
                                      _ptr(_f32(W1)), _ptr(_f32(b1)), _ptr(_f32(W2)), _ptr(_f32(b2)), float(c0), float(c1), int(mode), _ptr(out)))
        return out

    def op_ff(self, X, W1, b1, W2, b2, R1=None, c0=1.0, c1=1.0, fused=True):
        X = _f32(X); M, Cc = X.shape
        r = None if R1 is None else _f32(R1)
        out = np.empty((M, Cc), np.float32)
        self._ck(self.lib.ug_op_ff(self.ctx, _ptr(X), M, Cc, _ptr(_f32(W1)), _ptr(_f32(b1)), _ptr(_f32(W2)), _ptr(_f32(b2)), _ptr(r),
                                   float(c0), float(c1), int(bool(fused)), _ptr(out)))
        return out

    def set_fp8_linears(self, on=True):
        """MX-fp8 matrix instructions for the UNet transformers' linear layers (BASELINE configs[4]); reduced precision, default off."""
        self._ck(self.lib.ug_set_fp8_linears(self.ctx, int(bool(on))))

    def op_linear_mx8(self, A, W, bias=None, geglu=False, return_quant=False):
        A, W = _f32(A), _f32(W); M, K = A.shape; N = W.shape[0]
        b = None if bias is None else _f32(bias)
        out = np.empty((M, N // 2 if geglu else N), np.float32)
        a8 = np.empty((M, K), np.uint8) if return_quant else None
        sa = np.empty((K // 128, (M + 255) // 256 * 256), np.uint32) if return_quant else None
        self._ck(self.lib.ug_op_linear_mx8(self.ctx, _ptr(A), M, K, _ptr(W), N, _ptr(b), int(bool(geglu)), _ptr(out), _ptr(a8), _ptr(sa)))
        return (out, a8, sa) if return_quant else out

    def run_traced(self, steps, decode_chunk=8, with_normals=False):
        """ug_dc_run with the latents after every Euler step copied out: returns [steps, T, 4, h, w] float32."""
        T, H, W = self._shape
        tr = np.zeros((int(steps), T, H // 8, W // 8, 4), np.float32)
        self._ck(self.lib.ug_dc_set_trace(self.ctx, _ptr(tr), int(steps)))
        try:
            self.run(steps, decode_chunk, with_normals)
        finally:
            self.lib.ug_dc_set_trace(self.ctx, None, 0)
        return tr.transpose(0, 1, 4, 2, 3)

    def get_outputs(self, frames=True, depth=True, normals=False):
        T, H, W = self._shape
        fo = np.empty((T, H, W, 3), np.float32) if frames else None
        do = np.empty((T, H, W), np.float32) if depth else None
        no = np.empty((T, H, W, 3), np.float32) if normals else None
        self._ck(self.lib.ug_dc_get_outputs(self.ctx, _ptr(fo), _ptr(do), _ptr(no)))
        return fo, do, no

    def device_ptrs(self):
        """(frames, depth, normals) device addresses of the resident outputs + their shapes."""
        f, d, n = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._ck(self.lib.ug_dc_device_ptrs(self.ctx, C.byref(f), C.byref(d), C.byref(n)))
        T, H, W = self._shape
        return {"frames": (f.value, (T, H, W, 3)), "depth": (d.value, (T, H, W)), "normals": (n.value, (T, H, W, 3))}

    # ---- stages
    def clip_embed(self, frames):
        f = _f32(frames); T, H, W, _ = f.shape
        out = np.empty((T, self.clip_cfg.projection_dim), np.float32)
        self._ck(self.lib.ug_clip_embed(self.ctx, _ptr(f), T, H, W, _ptr(out)))
        return out

    def vae_encode(self, video_m11_thwc):
        f = _f32(video_m11_thwc); T, H, W, _ = f.shape
        out = np.empty((T, self.vae_cfg.latent_channels, H // 8, W // 8), np.float32)
        self._ck(self.lib.ug_vae_encode(self.ctx, _ptr(f), T, H, W, _ptr(out)))
        return out

    def vae_decode(self, z_tchw):
        z = _f32(z_tchw); T, _, h, w = z.shape
        out = np.empty((T, h * 8, w * 8, 3), np.float32)
        self._ck(self.lib.ug_vae_decode(self.ctx, _ptr(z), T, h, w, _ptr(out)))
        return out

    def unet_forward(self, sample_tchw, timestep, clip_emb):
        s = _f32(sample_tchw); T, _, h, w = s.shape
        e = _f32(clip_emb)
        out = np.empty((T, self.unet_cfg.out_channels, h, w), np.float32)
        self._ck(self.lib.ug_unet_forward(self.ctx, _ptr(s), T, h, w, float(timestep), _ptr(e), _ptr(out)))
        return out

    def normals_from_depth(self, depth, intrinsics):
        d = _f32(depth); T, H, W = d.shape
        k = _f32(intrinsics).reshape(T, 3, 3)
        out = np.empty((T, H, W, 3), np.float32)
        self._ck(self.lib.ug_normals_from_depth(self.ctx, _ptr(d), _ptr(k), T, H, W, _ptr(out)))
        return out

    # ---- metrics on device (pred=None: use the resident outputs of the last run)
    DEPTH_KEYS = ["Abs Rel", "Sq Rel", "RMSE", "Log RMSE", "delta < 1.", "delta < 1.25", "delta < 1.25^2", "delta < 1.25^3"]
    NORMAL_KEYS = ["normal mean", "normal median", "normal rmse", "angle < 5", "angle < 7.5", "angle < 11.25", "angle < 22.5", "angle < 30"]

    def eval_depth(self, gt, mask=None, pred=None, max_depth=80.0):
        g = _f32(gt); n = g.size
        p = None if pred is None else _f32(pred)
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
        out = np.zeros(11, np.float64)
        self._ck(self.lib.ug_eval_depth(self.ctx, _ptr(p), _ptr(g), _ptr(m), n, float(max_depth), _ptr(out)))
        res = dict(zip(self.DEPTH_KEYS, out[:8].tolist()))
        res["valid_pixels"] = int(out[8])
        return res, (float(out[9]), float(out[10]))

    def eval_normal(self, gt, mask=None, pred=None):
        g = _f32(gt); n = g.size // 3
        p = None if pred is None else _f32(pred)
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask).astype(np.uint8))
        out = np.zeros(8, np.float64)
        self._ck(self.lib.ug_eval_normal(self.ctx, _ptr(p), _ptr(g), _ptr(m), n, _ptr(out)))
        return dict(zip(self.NORMAL_KEYS, out.tolist()))

    # ---- ops (parity tests)
    def op_linear(self, A, W, bias=None, R1=None, c0=1.0, c1=1.0, act=0, geglu=False):
        A, W = _f32(A), _f32(W); M, K = A.shape; N = W.shape[0]
        b = None if bias is None else _f32(bias); r = None if R1 is None else _f32(R1)
        out = np.empty((M, N // 2 if geglu else N), np.float32)
        self._ck(self.lib.ug_op_linear(self.ctx, _ptr(A), M, K, _ptr(W), N, _ptr(b), _ptr(r), c0, c1, act, int(geglu), _ptr(out)))
        return out

    def op_conv(self, x0, weight, bias=None, x1=None, kt=1, k=3, stride=1, pad_t=1, pad_l=1, ups=1):
        x0 = _f32(x0); T, H, W, C0 = x0.shape
        x1a = None if x1 is None else _f32(x1); C1 = 0 if x1 is None else x1a.shape[-1]
        w = _f32(weight); O = w.shape[0]
        b = None if bias is None else _f32(bias)
        Ho, Wo = H * ups // stride, W * ups // stride
        out = np.empty((T, Ho, Wo, O), np.float32)
        self._ck(self.lib.ug_op_conv(self.ctx, _ptr(x0), C0, _ptr(x1a), C1, T, H, W, _ptr(w), _ptr(b), O, kt, k,
                                     stride, pad_t, pad_l, ups, _ptr(out)))
        return out

    def op_groupnorm(self, x0, G, eps, gamma, beta, x1=None, temporal=False, silu=False):
        x0 = _f32(x0); T, HW, C0 = x0.shape
        x1a = None if x1 is None else _f32(x1); C1 = 0 if x1 is None else x1a.shape[-1]
        out = np.empty((T, HW, C0 + C1), np.float32)
        self._ck(self.lib.ug_op_groupnorm(self.ctx, _ptr(x0), C0, _ptr(x1a), C1, T, HW, G, eps, int(temporal), int(silu),
                                          _ptr(_f32(gamma)), _ptr(_f32(beta)), _ptr(out)))
        return out

    def op_conv_gn(self, x, w, b, G, eps, gamma, beta, res=None, kt=1, k=3, temporal=False):
        """conv (+ residual) -> GroupNorm + SiLU with the statistics from a pass over the tensor / from the convolution's epilogue:
        (conv_out, y_pass, y_epi, rows_per_statistics_block)."""
        x = _f32(x); T, H, W, C0 = x.shape; w = _f32(w); O = w.shape[0]
        r = None if res is None else _f32(res)
        co = np.empty((T, H, W, O), np.float32); y1 = np.empty_like(co); y2 = np.empty_like(co)
        rb = C.c_int(0)
        self._ck(self.lib.ug_op_conv_gn(self.ctx, _ptr(x), C0, T, H, W, _ptr(w), _ptr(None if b is None else _f32(b)), _ptr(r), O, kt, k, G, eps, int(temporal),
                                        _ptr(_f32(gamma)), _ptr(_f32(beta)), _ptr(co), _ptr(y1), _ptr(y2), C.byref(rb)))
        return co, y1, y2, rb.value

    def op_layernorm(self, x, eps, gamma, beta, addvec=None, rows_per_vec=1):
        x = _f32(x); M, Cc = x.shape
        out = np.empty((M, Cc), np.float32); xout = np.empty((M, Cc), np.float32)
        av = None if addvec is None else _f32(addvec)
        self._ck(self.lib.ug_op_layernorm(self.ctx, _ptr(x), M, Cc, eps, _ptr(_f32(gamma)), _ptr(_f32(beta)), _ptr(av),
                                          rows_per_vec, _ptr(out), _ptr(xout)))
        return (out, xout) if addvec is not None else out

    def op_flash_attn(self, qkv, B, H, S):
        q = _f32(qkv); out = np.empty((B * S, H * 64), np.float32)
        self._ck(self.lib.ug_op_flash_attn(self.ctx, _ptr(q), B, H, S, _ptr(out)))
        return out

    def op_temporal_attn(self, qkv, T, HW, H):
        q = _f32(qkv); out = np.empty((T * HW, H * 64), np.float32)
        self._ck(self.lib.ug_op_temporal_attn(self.ctx, _ptr(q), T, HW, H, _ptr(out)))
        return out

    def op_attention_generic(self, qkv, B, S, H, d):
        q = _f32(qkv); out = np.empty((B * S, H * d), np.float32)
        self._ck(self.lib.ug_op_attention_generic(self.ctx, _ptr(q), B, S, H, d, _ptr(out)))
        return out

    def op_flash_attn_dh(self, qkv, B, S, H, d):
        q = _f32(qkv); out = np.empty((B * S, H * d), np.float32)
        self._ck(self.lib.ug_op_flash_attn_dh(self.ctx, _ptr(q), B, S, H, d, _ptr(out)))
        return out

    def op_euler_step(self, v, lat, sigma, sigma_next):
        v = _f32(v); l = _f32(lat).copy()
        self._ck(self.lib.ug_op_euler_step(self.ctx, _ptr(v), _ptr(l), l.size, sigma, sigma_next))
        return l

    def bench_gemm(self, M=0, N=0, K=0, conv=None, cfg=-1, split=0, iters=20):
        """conv = dict(T,H,W,C0,C1,kt,k,stride,ups) or None (dense).  Returns (ms, TFLOP/s, cfg, split)."""
        out = np.zeros(8, np.float32)
        cv = conv or {}
        self._ck(self.lib.ug_bench_gemm(self.ctx, M, N, K, int(conv is not None), cv.get("T", 0), cv.get("H", 0), cv.get("W", 0),
                                        cv.get("C0", 0), cv.get("C1", 0), cv.get("kt", 1), cv.get("k", 1), cv.get("stride", 1),
                                        cv.get("ups", 1), cfg, split, iters, _ptr(out)))
        Mr, Kr = float(out[3]), float(out[4])
        return float(out[0]), 2.0 * Mr * N * Kr / (out[0] * 1e-3) / 1e12, int(out[1]), int(out[2])

    def bench_mfma_peak(self, iters=20000):
        """Calibration: chip-wide fp16 MFMA TFLOP/s with operands in registers (kernels/probe.hip)."""
        out = np.zeros(1, np.float32)
        self.lib.ug_bench_mfma_peak.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self._ck(self.lib.ug_bench_mfma_peak(self.ctx, int(iters), _ptr(out)))
        return float(out[0])

    def tune_force(self, cfg=-1, split=-1):
        """Test / A-B aid, per context: force a GEMM tile config + split-K factor ((-1, -1) = planner); cfg = -100 - mask sets the knob mask."""
        self._ck(self.lib.ug_tune_force(self.ctx, int(cfg), int(split)))

    def bench_groupnorm(self, C0, C1, T, HW, temporal, mode, iters=20):
        out = np.zeros(1, np.float32)
        self._ck(self.lib.ug_bench_groupnorm(self.ctx, C0, C1, T, HW, int(temporal), mode, iters, _ptr(out)))
        return float(out[0])

    # ---- profiling
    def profile_begin(self, shapes=False):
        self._ck((self.lib.ug_profile_begin_shapes if shapes else self.lib.ug_profile_begin)(self.ctx))

    def profile_end(self):
        prof = json.loads(self.lib.ug_profile_end(self.ctx).decode())
        self.last_profile_order = prof.pop("__order__", None)      # shapes=True: [[name, algorithmic bytes], ...] of the GEMM launches, in order
        return prof

    def workspace_peak(self):
        return int(self.lib.ug_workspace_peak(self.ctx))
