"""CPU: structural known-answers that pin the oracle / the product's weight manifests."""
import dataclasses
import math

import numpy as np
import pytest
import torch

from unigeo_amd import weights as W


def _sd_shapes(mod):
    return {k: tuple(v.shape) for k, v in mod.state_dict().items()}


def test_manifests_match_oracle_modules_and_param_counts():
    from oracle.clip import CLIPConfig, CLIPVisionWithProjection
    from oracle.svd_unet import UNetConfig, UNetSpatioTemporal
    from oracle.vae import AutoencoderKLTemporalDecoder, VAEConfig
    with torch.device("meta"):
        u, v, c = UNetSpatioTemporal(), AutoencoderKLTemporalDecoder(), CLIPVisionWithProjection()
        assert _sd_shapes(u) == {k: tuple(s) for k, s in W.unet_manifest().items()}
        assert _sd_shapes(v) == {k: tuple(s) for k, s in W.vae_manifest().items()}
        assert _sd_shapes(c) == {k: tuple(s) for k, s in W.clip_manifest().items()}
        tu, tv, tc = W.tiny_cfgs()
        assert _sd_shapes(UNetSpatioTemporal(UNetConfig(**dataclasses.asdict(tu)))) == {k: tuple(s) for k, s in W.unet_manifest(tu).items()}
        assert _sd_shapes(AutoencoderKLTemporalDecoder(VAEConfig(**dataclasses.asdict(tv)))) == {k: tuple(s) for k, s in W.vae_manifest(tv).items()}
        assert _sd_shapes(CLIPVisionWithProjection(CLIPConfig(**dataclasses.asdict(tc)))) == {k: tuple(s) for k, s in W.clip_manifest(tc).items()}
    n = lambda m: sum(int(np.prod(s)) for s in m.values())
    assert n(W.unet_manifest()) == 1_524_623_082          # SVD-XT UNet: 1.52 B parameters (known answer)
    assert n(W.clip_manifest()) == 632_076_800            # CLIP ViT-H/14 vision tower + projection
    assert n(W.vae_manifest()) == 97_742_847


def test_clip_oracle_matches_transformers():
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    from oracle.clip import CLIPConfig, CLIPVisionWithProjection
    _, _, c = W.tiny_cfgs()
    tc = CLIPVisionConfig(hidden_size=c.hidden_size, intermediate_size=c.intermediate_size,
                          num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                          image_size=c.image_size, patch_size=c.patch_size, projection_dim=c.projection_dim, hidden_act="gelu")
    torch.manual_seed(0)
    ref = CLIPVisionModelWithProjection(tc).eval()
    mine = CLIPVisionWithProjection(CLIPConfig(**dataclasses.asdict(c))).eval()
    mine.load_state_dict({k: v for k, v in ref.state_dict().items() if "position_ids" not in k})
    x = torch.randn(2, 3, 224, 224)
    with torch.no_grad():
        a, b = ref(pixel_values=x).image_embeds, mine(x)
    assert float((a - b).abs().max()) < 1e-5


def _hf_dino(hidden, inter, layers, heads, image_size, seed=0):
    from transformers import Dinov2Config, Dinov2Model
    tc = Dinov2Config(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, mlp_ratio=inter // hidden, image_size=image_size, patch_size=14,
                      layer_norm_eps=1e-6, hidden_act="gelu", layerscale_value=1.0, use_swiglu_ffn=False, qkv_bias=True)
    torch.manual_seed(seed)
    ref = Dinov2Model(tc).eval()
    with torch.no_grad():                                     # position table / class token / LayerScale are constant-initialised: make every one of them matter
        for n_, p_ in ref.named_parameters():
            if "position_embeddings" in n_ or "cls_token" in n_:
                p_.copy_(torch.randn_like(p_) * 0.5)
            if "lambda1" in n_:
                p_.copy_(1.0 + 0.3 * torch.randn_like(p_))
    return ref


def test_dinov2_oracle_matches_transformers():
    """VERDICT r5 missing 3: oracle.stablenormal.DinoV2 (the DINO tower of the StableNormal restatement; reference call site model/stablenormal.py:16,39) pinned to
    `transformers.Dinov2Model` with shared random weights, through the state-dict conversion the checkpoint loader uses (weights.dinov2_hf_to_hub) - same recipe
    as the CLIP tower above.  Also at the real ViT-L/14 widths with 2 layers (head_dim 64, LayerScale, 16 x 16 grid)."""
    from oracle.stablenormal import DinoConfig, DinoV2
    for hidden, inter, layers, heads, size in ((64, 128, 2, 1, 224), (1024, 4096, 2, 16, 224)):
        ref = _hf_dino(hidden, inter, layers, heads, size)
        st = W.dinov2_hf_to_hub({k: v.numpy() for k, v in ref.state_dict().items()})
        cfg = W.DinoCfg(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads, image_size=size)
        st = {k: v for k, v in st.items() if k != "mask_token"}
        W.check_against_manifest(st, W.dino_manifest(cfg), "dino")        # the converted checkpoint is exactly what the engine's manifest lists
        mine = DinoV2(DinoConfig(**dataclasses.asdict(cfg))).eval()
        mine.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
        x = torch.randn(2, 3, size, size)
        with torch.no_grad():
            a, b = ref(pixel_values=x).last_hidden_state[:, 1:], mine(x)
        err = float((a - b).abs().max() / a.abs().max())
        assert a.shape == b.shape and err < 2e-5, (hidden, err)


def test_dinov2_position_table_resampling_matches_transformers():
    """A real DINOv2 checkpoint holds a 37 x 37 position table (518 / 14); the tower runs on a 224 x 224 copy of the image = a 16 x 16 grid.  The loader resamples
    the table ONCE (weights.resample_dino_pos_embed); transformers resamples it on every call (Dinov2Embeddings.interpolate_pos_encoding): a Dinov2Model built for
    518 x 518 and fed 224 x 224 images must equal the oracle tower built for 224 with the loader's resampled table."""
    from oracle.stablenormal import DinoConfig, DinoV2
    ref = _hf_dino(64, 128, 2, 1, 518, seed=1)
    st = W.dinov2_hf_to_hub({k: v.numpy() for k, v in ref.state_dict().items()})
    assert st["pos_embed"].shape == (1, 1 + 37 * 37, 64)
    st["pos_embed"] = W.resample_dino_pos_embed(st["pos_embed"], 16)
    cfg = W.DinoCfg(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1, image_size=224)
    W.check_against_manifest({k: v for k, v in st.items() if k != "mask_token"}, W.dino_manifest(cfg), "dino")
    mine = DinoV2(DinoConfig(**dataclasses.asdict(cfg))).eval()
    mine.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items() if k != "mask_token"})
    x = torch.randn(2, 3, 224, 224)
    with torch.no_grad():
        a, b = ref(pixel_values=x).last_hidden_state[:, 1:], mine(x)
    assert float((a - b).abs().max() / a.abs().max()) < 2e-5


def test_scheduler_closed_form_tables():
    from oracle.scheduler import EulerKarrasVPred
    for n in (2, 5, 25):
        s = EulerKarrasVPred(); ts = s.set_timesteps(n)
        sig = s.sigmas.numpy()
        assert sig[0] == pytest.approx(700.0, rel=1e-6) and sig[n - 1] == pytest.approx(0.002, rel=1e-5) and sig[n] == 0
        assert np.all(np.diff(sig) < 0)
        np.testing.assert_allclose(ts.numpy(), 0.25 * np.log(sig[:n]), rtol=1e-6)
        assert s.init_noise_sigma == pytest.approx(math.sqrt(700.0 ** 2 + 1), rel=1e-6)
    # an exact denoiser (v-pred of a pure-noise sample around x0 = 0) is driven to 0 by the Euler steps
    s = EulerKarrasVPred(); s.set_timesteps(5)
    x = torch.randn(1000) * s.init_noise_sigma
    for i in range(5):
        sg = s.sigmas[i]
        v = x / (sg * (sg ** 2 + 1) ** 0.5)                     # x0 = -s/sqrt(s^2+1)*v + x/(s^2+1) = 0
        x = s.step(v, i, x)
    assert float(x.abs().max()) < 1e-3


def test_weight_loader_rejects_wrong_architecture():
    u, _, _ = W.tiny_cfgs()
    st = W.random_state(W.unet_manifest(u), 0)
    W.check_against_manifest(st, W.unet_manifest(u), "unet")
    bad = dict(st); bad.pop("conv_in.weight"); bad["surprise.weight"] = np.zeros(3, np.float16)
    with pytest.raises(ValueError):
        W.check_against_manifest(bad, W.unet_manifest(u), "unet")
    bad = dict(st); bad["conv_in.weight"] = np.zeros((1, 2, 3, 3), np.float16)
    with pytest.raises(ValueError):
        W.check_against_manifest(bad, W.unet_manifest(u), "unet")


def test_stablenormal_restatement_structure():
    """Known answers for the StableNormal restatement (un-vendored hub repo, PARITY UNPINNED): the published parameter counts of the
    SD 2.1 UNet2DConditionModel (865 910 724) and the SD AutoencoderKL (83 653 863), the ControlNet trunk, DINOv2 ViT-L/14 at a
    16x16 grid; the engine's manifests (what real safetensors are checked against) list exactly the oracle's state-dict keys/shapes."""
    import dataclasses
    from oracle.stablenormal import AutoencoderKL, ControlNet, DinoV2, SDUNet
    from unigeo_amd import weights as W
    mods = {"unet": (SDUNet(), W.sd_unet_manifest(), 865_910_724), "controlnet": (ControlNet(), W.controlnet_manifest(), 363_141_760),
            "controlnet_dino": (ControlNet(dino_dim=1024), W.controlnet_manifest(dino_dim=1024), 363_141_760 + 1024 * 320 + 320),
            "vae": (AutoencoderKL(), W.sd_vae_manifest(), 83_653_863), "dino": (DinoV2(), W.dino_manifest(), 303_227_904)}
    for name, (m, man, count) in mods.items():
        sd = m.state_dict()
        assert sum(v.numel() for v in sd.values()) == count, name
        assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v) for k, v in man.items()}, name


def test_stablenormal_ddim_tables_and_plugin_refusal():
    import numpy as np
    import pytest
    from unigeo_amd.stablenormal import ddim_tables, normals_to_uint8, refine_timesteps
    from oracle.stablenormal import ddim_tables as o_tables
    ts = refine_timesteps(401, 10)
    assert ts[0] == 401 and len(ts) == 10 and all(a > b for a, b in zip(ts, ts[1:]))
    for pt in ("epsilon", "v_prediction", "sample"):
        a, b = ddim_tables(ts, pt); a2, b2 = o_tables(ts, pt)
        assert np.array_equal(a, a2) and np.array_equal(b, b2)
    # closed form: the last step lands on alpha_bar_prev = 1, i.e. x <- x0;  for sample prediction that is a = 0, b = 1
    a, b = ddim_tables(ts, "sample")
    assert abs(a[-1]) < 1e-7 and abs(b[-1] - 1) < 1e-7
    assert normals_to_uint8(np.array([[-1.0, 0.0, 1.0]]))[0].tolist() == [0, 127, 255]
    from unigeo_amd.model import StableNormal
    with pytest.raises(FileNotFoundError):
        StableNormal(model_dir="/nonexistent/dir")


def test_stablenormal_checkpoint_directory_round_trip(tmp_path):
    """load_stablenormal_pretrained on a tiny checkpoint directory in the documented layout (S11): every component is checked against
    its manifest, DINO's larger position table is resampled to the tower's grid, a stray tensor is refused with a precise message."""
    import numpy as np
    import pytest
    from safetensors.numpy import save_file
    from unigeo_amd import weights as W
    from unigeo_amd.stablenormal import COMPONENTS, manifests
    cfgs = W.tiny_sn_cfgs()
    ms = manifests(cfgs)
    states = {c: W.random_state(ms[c], 3 + i) for i, c in enumerate(COMPONENTS)}
    d = cfgs[2].hidden_size
    big = np.random.default_rng(0).standard_normal((1, 1 + 37 * 37, d)).astype(np.float16)      # a 518 / 14 = 37 x 37 table, as dinov2 ships
    for c in COMPONENTS:
        (tmp_path / c).mkdir()
        st = dict(states[c])
        if c == "dino":
            st["pos_embed"] = big; st["mask_token"] = np.zeros((1, d), np.float16)
        save_file(st, str(tmp_path / c / ("model.safetensors" if c == "dino" else "diffusion_pytorch_model.fp16.safetensors")))
    pe = np.random.default_rng(1).standard_normal((77, cfgs[0].cross_attention_dim)).astype(np.float32)
    np.save(tmp_path / "prompt_embeds.npy", pe)
    got, prompt = W.load_stablenormal_pretrained(str(tmp_path), cfgs)
    assert np.array_equal(prompt, pe)
    for c in COMPONENTS:
        assert set(got[c]) == set(ms[c]) and all(tuple(got[c][k].shape) == tuple(v) for k, v in ms[c].items()), c
    assert got["dino"]["pos_embed"].shape == (1, 257, d) and np.array_equal(got["dino"]["pos_embed"][:, 0], big[:, 0].astype(np.float32))
    assert np.array_equal(got["unet"]["conv_in.weight"], states["unet"]["conv_in.weight"])
    # the prompt-embedding path without a precomputed file (VERDICT r5 missing 3): `text_encoder/` + `tokenizer/` in transformers' own format - the loader runs
    # transformers.CLIPTextModel on the fixed prompt "The normal map" (77 tokens, padded) and hands over last_hidden_state; nothing of it is restated, so the pin is
    # that the loader's branch returns exactly what transformers returns for that prompt (reference call site: model/stablenormal.py:16, the hub pipeline's prompt)
    import json
    import torch
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer
    (tmp_path / "prompt_embeds.npy").unlink()
    (tmp_path / "tokenizer").mkdir()
    chars = sorted(set("thenormalmap"))
    vocab = {c: i for i, c in enumerate(chars)}
    vocab.update({c + "</w>": len(chars) + i for i, c in enumerate(chars)})
    vocab["<|startoftext|>"] = len(vocab); vocab["<|endoftext|>"] = len(vocab)
    (tmp_path / "tokenizer" / "vocab.json").write_text(json.dumps(vocab)); (tmp_path / "tokenizer" / "merges.txt").write_text("#version: 0.2\n")
    CLIPTokenizer(str(tmp_path / "tokenizer" / "vocab.json"), str(tmp_path / "tokenizer" / "merges.txt")).save_pretrained(str(tmp_path / "tokenizer"))
    torch.manual_seed(5)
    enc = CLIPTextModel(CLIPTextConfig(vocab_size=len(vocab), hidden_size=cfgs[0].cross_attention_dim, intermediate_size=96, num_hidden_layers=2, num_attention_heads=2,
                                       max_position_embeddings=77, bos_token_id=vocab["<|startoftext|>"], eos_token_id=vocab["<|endoftext|>"])).eval()
    enc.save_pretrained(str(tmp_path / "text_encoder"))
    _, prompt2 = W.load_stablenormal_pretrained(str(tmp_path), cfgs)
    ids = CLIPTokenizer.from_pretrained(str(tmp_path / "tokenizer"))("The normal map", padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
    assert int(ids[0, 0]) == vocab["<|startoftext|>"] and int(ids[0, -1]) == vocab["<|endoftext|>"] and ids.shape == (1, 77)
    with torch.no_grad():
        want = enc(ids).last_hidden_state[0].numpy()
    assert prompt2.shape == (77, cfgs[0].cross_attention_dim) and np.array_equal(prompt2, want)
    np.save(tmp_path / "prompt_embeds.npy", pe)
    st = dict(states["unet"]); st["surprise.weight"] = np.zeros((2, 2), np.float16)
    save_file(st, str(tmp_path / "unet" / "diffusion_pytorch_model.fp16.safetensors"))
    with pytest.raises(ValueError, match="unexpected 1"):
        W.load_stablenormal_pretrained(str(tmp_path), cfgs)
