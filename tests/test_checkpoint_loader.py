"""Row a1: the construction path of the reference plugin (``/root/reference/model/depthcrafter.py:18-29``: two ``from_pretrained`` calls
on the diffusers directory layout, ``variant="fp16"`` for the SVD components, an fp32 UNet file) on a tiny-configuration checkpoint
directory written by ``weights.save_pretrained_layout``.

CPU part: the loader reads the three ``config.json`` files and ``scheduler/scheduler_config.json``, derives the architecture from them,
checks every tensor against that architecture's manifest and hard-fails on any disagreement.
GPU part: ``DepthCrafter(model_dir, unet_path, pre_train_path)`` == the same weights bound through ``from_state``, bit for bit.
"""
import dataclasses
import json
import os

import numpy as np
import pytest

from unigeo_amd import weights as W


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    root = tmp_path_factory.mktemp("ckpt")
    cfgs = W.tiny_cfgs()
    u, v, c = cfgs
    states = (W.random_state(W.unet_manifest(u), 1), W.random_state(W.vae_manifest(v), 2), W.random_state(W.clip_manifest(c), 3))
    unet_path, pre = str(root / "DepthCrafter"), str(root / "stable-video-diffusion-img2vid-xt")
    W.save_pretrained_layout(unet_path, pre, *states, cfgs=cfgs)
    return dict(unet_path=unet_path, pre=pre, cfgs=cfgs, states=states)


def test_layout_is_the_diffusers_one(ckpt):
    for rel in ("config.json", "diffusion_pytorch_model.safetensors"):
        assert os.path.exists(os.path.join(ckpt["unet_path"], rel))
    for rel in ("vae/config.json", "vae/diffusion_pytorch_model.fp16.safetensors", "image_encoder/config.json",
                "image_encoder/model.fp16.safetensors", "scheduler/scheduler_config.json"):
        assert os.path.exists(os.path.join(ckpt["pre"], rel)), rel
    from safetensors.numpy import load_file
    u = load_file(os.path.join(ckpt["unet_path"], "diffusion_pytorch_model.safetensors"))
    assert next(iter(u.values())).dtype == np.float32          # the DepthCrafter UNet release is fp32; the reference casts with torch_dtype=fp16


def test_loader_reads_configs_and_round_trips(ckpt):
    u, v, c, cfgs = W.load_pretrained(ckpt["unet_path"], ckpt["pre"], with_cfgs=True)
    for got, want in zip(cfgs, ckpt["cfgs"]):
        assert dataclasses.asdict(got) == dataclasses.asdict(want)
    for got, want in zip((u, v, c), ckpt["states"]):
        assert set(got) == set(want)
        for k in want:
            assert np.array_equal(np.asarray(got[k], np.float32), np.asarray(want[k], np.float32)), k   # fp16 -> fp32 file -> exact


def test_default_config_files_give_the_restated_architecture(tmp_path):
    """The keys of the real SVD-XT / DepthCrafter config files map onto the default (full-size) configuration objects."""
    ucfg = W.unet_cfg_from_config({"block_out_channels": [320, 640, 1280, 1280], "num_attention_heads": [5, 10, 20, 20], "in_channels": 8, "out_channels": 4,
                                   "down_block_types": ["CrossAttnDownBlockSpatioTemporal"] * 3 + ["DownBlockSpatioTemporal"],
                                   "up_block_types": ["UpBlockSpatioTemporal"] + ["CrossAttnUpBlockSpatioTemporal"] * 3, "layers_per_block": 2,
                                   "cross_attention_dim": 1024, "addition_time_embed_dim": 256, "projection_class_embeddings_input_dim": 768,
                                   "transformer_layers_per_block": 1, "num_frames": 25, "sample_size": 96})
    assert dataclasses.asdict(ucfg) == dataclasses.asdict(W.UNetCfg())
    vcfg = W.vae_cfg_from_config({"_class_name": "AutoencoderKLTemporalDecoder", "block_out_channels": [128, 256, 512, 512], "latent_channels": 4,
                                  "layers_per_block": 2, "scaling_factor": 0.18215, "force_upcast": True, "in_channels": 3, "out_channels": 3})
    assert dataclasses.asdict(vcfg) == dataclasses.asdict(W.VAECfg())
    ccfg = W.clip_cfg_from_config({"hidden_act": "gelu", "hidden_size": 1280, "intermediate_size": 5120, "num_hidden_layers": 32, "num_attention_heads": 16,
                                   "image_size": 224, "patch_size": 14, "projection_dim": 1024, "layer_norm_eps": 1e-5})
    assert dataclasses.asdict(ccfg) == dataclasses.asdict(W.CLIPCfg())
    W.check_scheduler_config({"_class_name": "EulerDiscreteScheduler", "beta_end": 0.012, "beta_schedule": "scaled_linear", "beta_start": 0.00085,
                              "interpolation_type": "linear", "num_train_timesteps": 1000, "prediction_type": "v_prediction", "sigma_max": 700.0,
                              "sigma_min": 0.002, "steps_offset": 1, "timestep_spacing": "leading", "timestep_type": "continuous", "use_karras_sigmas": True})


def _edit(path, **kv):
    with open(path) as f:
        d = json.load(f)
    d.update(kv)
    with open(path, "w") as f:
        json.dump(d, f)


@pytest.mark.parametrize("rel,key,bad,exc", [
    ("scheduler/scheduler_config.json", "sigma_max", 80.0, ValueError),
    ("scheduler/scheduler_config.json", "prediction_type", "epsilon", ValueError),
    ("scheduler/scheduler_config.json", "timestep_spacing", "trailing", ValueError),
    ("scheduler/scheduler_config.json", "use_karras_sigmas", False, ValueError),
    ("vae/config.json", "force_upcast", False, ValueError),
    ("vae/config.json", "block_out_channels", [32, 64, 64, 32], ValueError),      # tensors no longer match the architecture the file claims
    ("image_encoder/config.json", "hidden_act", "quick_gelu", ValueError),
    ("image_encoder/config.json", "num_hidden_layers", 3, ValueError),
])
def test_loader_hard_fails_on_disagreement(ckpt, tmp_path, rel, key, bad, exc):
    import shutil
    pre = str(tmp_path / "pre")
    shutil.copytree(ckpt["pre"], pre)
    _edit(os.path.join(pre, rel), **{key: bad})
    with pytest.raises(exc):
        W.load_pretrained(ckpt["unet_path"], pre)


def test_loader_hard_fails_on_unet_config_and_missing_files(ckpt, tmp_path):
    import shutil
    up = str(tmp_path / "unet")
    shutil.copytree(ckpt["unet_path"], up)
    _edit(os.path.join(up, "config.json"), num_attention_heads=[2, 2, 2, 2])          # head dim != 64
    with pytest.raises(ValueError):
        W.load_pretrained(up, ckpt["pre"])
    _edit(os.path.join(up, "config.json"), num_attention_heads=[1, 2, 2, 2], transformer_layers_per_block=2)
    with pytest.raises(ValueError):
        W.load_pretrained(up, ckpt["pre"])
    os.remove(os.path.join(up, "config.json"))
    with pytest.raises(FileNotFoundError):
        W.load_pretrained(up, ckpt["pre"])


@pytest.mark.gpu
def test_plugin_from_checkpoint_directory_equals_from_state(ckpt):
    """DepthCrafter(model_dir, unet_path, pre_train_path) (reference constructor signature, configs/depthcrafter_scannetpp.yaml:10-13)
    == the same tensors through from_state: bit-identical depth and normals."""
    from unigeo_amd.model.depthcrafter import DepthCrafter
    from unigeo_amd.pipeline import DepthCrafterPipelineHIP
    from unigeo_amd.synthetic import synthetic_clip
    data = synthetic_clip(3, 64, 64, seed=2)
    plug = DepthCrafter(model_dir="/unused", unet_path=ckpt["unet_path"], pre_train_path=ckpt["pre"], num_inference_steps=2, workspace_bytes=3 << 30)
    try:
        a = plug.forward(data)
    finally:
        plug.pipeline.engine.close()
    ref = DepthCrafter(synthetic_weights=True, cfgs=ckpt["cfgs"], num_inference_steps=2, workspace_bytes=3 << 30)
    ref.pipeline.engine.close()
    ref.pipeline = DepthCrafterPipelineHIP.from_state(*ckpt["states"], cfgs=ckpt["cfgs"], workspace_bytes=3 << 30)
    try:
        b = ref.forward(data)
    finally:
        ref.pipeline.engine.close()
    assert np.array_equal(a["pred_depths"].numpy(), b["pred_depths"].numpy())
    assert np.array_equal(a["pred_normals"].numpy(), b["pred_normals"].numpy())
