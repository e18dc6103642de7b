#!/usr/bin/env python
"""Pin the StableNormal restatement (uncertainty register S1 - S12) on a machine that HAS what the build container
lacks: network access for ``torch.hub.load("Stable-X/StableNormal", "StableNormal", trust_repo=True)`` (the call at
/root/reference/model/stablenormal.py:16) and the checkpoints it downloads.  NOT exercised by the test-suite.

What it does
  1. loads the hub predictor exactly as the reference does and runs it on one image (``predictor(pil) -> pil``, :39);
  2. walks the predictor object for the torch modules it is made of (VAE, UNets, ControlNets, DINO, text encoder) and hooks them, so the run also
     yields: the module class names + call counts in call order (S3 - S6, S8, S9: how many UNet / ControlNet evaluations, at which timesteps,
     with which sample shapes - which answers the processing-resolution question S1 directly), every scheduler call's timestep (S9 / S10)
     and the tensors at the stage boundaries the restatement names: image latent, YOSO latent, DINO tokens, per-step latents, decoder input;
  3. loads the SAME checkpoints into this repository's HIP predictor (``--model-dir`` = a directory in the layout of
     ``weights.load_stablenormal_pretrained``; ``tools/..`` prints the state-dict key sets that do not match, which pins S2 - S5 structurally),
     runs the same image with the knobs the hub run revealed (``--processing-resolution``, ``--prediction-type``, ``--refine-steps`` ...),
  4. prints, stage by stage, max |difference| / max |reference| and stops at the FIRST stage that diverges by more than ``--tol``.

    python tools/compare_with_stablenormal_hub.py --image some.png --model-dir /path/to/stablenormal_ckpts [--processing-resolution 768]
"""
import argparse
import collections
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def find_modules(obj, depth=0, seen=None, path="predictor"):
    """Every torch.nn.Module reachable through plain attributes of the hub predictor (it is a thin wrapper around a diffusers-style pipeline)."""
    import torch
    seen = seen if seen is not None else set()
    out = []
    if id(obj) in seen or depth > 4:
        return out
    seen.add(id(obj))
    if isinstance(obj, torch.nn.Module):
        return [(path, obj)]
    for name in dir(obj):
        if name.startswith("__"):
            continue
        try:
            v = getattr(obj, name)
        except Exception:
            continue
        if isinstance(v, torch.nn.Module):
            out.append((f"{path}.{name}", v))
        elif hasattr(v, "__dict__") and not callable(v):
            out += find_modules(v, depth + 1, seen, f"{path}.{name}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--image", required=True)
    ap.add_argument("--model-dir", default=None, help="checkpoint directory for the HIP predictor (weights.load_stablenormal_pretrained layout); omit to only trace the hub run")
    ap.add_argument("--processing-resolution", type=int, default=0)
    ap.add_argument("--prediction-type", default="v_prediction")
    ap.add_argument("--refine-steps", type=int, default=10)
    ap.add_argument("--refine-start", type=int, default=401)
    ap.add_argument("--yoso-timestep", type=int, default=999)
    ap.add_argument("--tol", type=float, default=2e-2)
    a = ap.parse_args()
    import torch
    from PIL import Image
    img = Image.open(a.image).convert("RGB")

    # ---- 1 / 2: the hub predictor, traced
    predictor = torch.hub.load("Stable-X/StableNormal", "StableNormal", trust_repo=True)
    trace, calls = [], collections.Counter()

    def hook(name):
        def fn(mod, inp, out):
            first = next((x for x in inp if torch.is_tensor(x)), None)
            o = out[0] if isinstance(out, (tuple, list)) else getattr(out, "sample", out)
            calls[name] += 1
            trace.append((name, type(mod).__name__, tuple(first.shape) if first is not None else None,
                          [float(x) if torch.is_tensor(x) and x.numel() == 1 else None for x in inp[1:2]],
                          o.detach().float().cpu().numpy() if torch.is_tensor(o) and o.numel() < 5_000_000 else None))
        return fn
    mods = find_modules(predictor)
    for name, m in mods:
        m.register_forward_hook(hook(name))
    print("modules found in the hub predictor:")
    for name, m in mods:
        print(f"  {name:50s} {type(m).__name__:40s} {sum(p.numel() for p in m.parameters()):>13,d} parameters")
    ref_pil = predictor(img)
    ref = np.asarray(ref_pil)
    print("\ncall order (module, class, first-argument shape, timestep):")
    for name, cls, shp, ts, _ in trace:
        print(f"  {name:50s} {cls:36s} {shp} t={ts}")
    print("\ncall counts:", dict(calls))
    print(f"hub output: {ref.shape} {ref.dtype}; input {img.size}: a first-argument shape other than (1, 4, H/8, W/8) of the INPUT size above means the hub "
          "predictor resizes to a processing resolution (S1)")
    if not a.model_dir:
        return

    # ---- 3: the HIP predictor on the same checkpoints
    from unigeo_amd.stablenormal import StableNormalPredictorHIP, normals_to_uint8
    pred = StableNormalPredictorHIP.from_pretrained(a.model_dir, processing_resolution=a.processing_resolution, prediction_type=a.prediction_type,
                                                    refine_steps=a.refine_steps, refine_start=a.refine_start, yoso_timestep=a.yoso_timestep)
    x = np.asarray(img, np.float32)[None] / 255.0
    H, W = x.shape[1:3]
    if not a.processing_resolution and (H % 64 or W % 64):
        raise SystemExit("image size is not a multiple of 64: pass --processing-resolution")
    got = normals_to_uint8(pred.predict_batch(x)[0])

    # ---- 4: stage by stage (the hub trace gives the reference tensors; the engine's stage entry points give ours)
    def rel(u, v):
        return float(np.abs(u - v).max() / (np.abs(v).max() + 1e-12))
    eng = pred.engine
    stages = []
    vae_out = [t for t in trace if "vae" in t[0].lower() and t[4] is not None and t[4].ndim == 4 and t[4].shape[1] in (4, 8)]
    if vae_out and not a.processing_resolution:
        z_ref = vae_out[0][4][:, :4]
        z_hip = eng.sn_vae_encode((x * 2 - 1).astype(np.float32))
        stages.append(("image latent (VAE encode, mode)", rel(z_hip, z_ref)))
    dino = [t for t in trace if "dino" in t[0].lower() and t[4] is not None and t[4].ndim == 3]
    if dino and not a.processing_resolution:
        tok_ref = dino[-1][4]
        tok_hip = eng.sn_dino(x)
        if tok_ref.shape[1] == tok_hip.shape[1] + 1:
            tok_ref = tok_ref[:, 1:]
        if tok_ref.shape == tok_hip.shape:
            stages.append(("DINO patch tokens", rel(tok_hip, tok_ref)))
        else:
            print(f"DINO token shapes differ: hub {tok_ref.shape} vs HIP {tok_hip.shape} (S5: input size / projection stem)")
    ang = np.degrees(np.arccos(np.clip(((got.astype(np.float32) / 127.5 - 1) * (ref.astype(np.float32) / 127.5 - 1)).sum(-1), -1, 1)))
    stages.append(("final normals (degrees, mean)", float(ang.mean()) / 90.0))
    print("\nstage differences (max |d| / max |ref|; final: mean angle / 90 deg):")
    for name, e in stages:
        flag = "   <-- FIRST DIVERGENCE" if e > a.tol else ""
        print(f"  {name:40s} {e:.3e}{flag}")
        if flag:
            break
    print(f"final normals: mean angle {ang.mean():.2f} deg, 99th percentile {np.percentile(ang, 99):.2f} deg")


if __name__ == "__main__":
    main()
