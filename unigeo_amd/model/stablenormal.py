"""StableNormal plugin on the MI355X-native engine.

Mirrors ``/root/reference/model/stablenormal.py``: ``__init__(**kwargs)`` (:9-18: prints the device, builds ``self.predictor``),
``prepare_input`` (:21-27), ``forward(data)`` (:30-52) with the exact uint8 post-processing (x flip on uint8 wraps mod 256 at :43,
``/255*2-1`` at :45, zero depths at :49) - pinned by golden G7.

The reference's predictor is ``torch.hub.load("Stable-X/StableNormal", ...)`` (:16): network access plus an un-vendored repository.
Here ``self.predictor`` is ``StableNormalPredictorHIP`` (unigeo_amd/stablenormal.py -> ``ug_sn_run``): same call
(``predictor(pil) -> pil``), restated architecture, PARITY UNPINNED.  Deliberate, visible differences:
  * the frames of a clip go through the GPU as ONE batch (the reference loops over them; the network is spatial-only, frames are
    independent, so the result per frame is the same);
  * checkpoints are read from ``model_dir`` (the reference ignores ``model_dir`` and downloads); without them the constructor raises
    unless ``synthetic_weights=True`` is passed;
  * a caller-supplied ``predictor=`` (any callable image -> uint8 normal image) is still honoured.
"""
import os
import warnings

import numpy as np

PARITY_NOTE = ("StableNormal on the MI355X engine is a RESTATEMENT of the published design (DESIGN.md section 9): it has never been run next to the "
               "torch.hub predictor the reference loads (model/stablenormal.py:16), so its normals are NOT known to be comparable with the "
               "reference's - parity unpinned.  Pass predictor=<callable> to drive the plugin with the hub predictor itself.")


class StableNormal:
    parity = "unpinned"

    def __init__(self, **kwargs):
        device_id = int(kwargs.get("device_id", 0))
        self.device = f"hip:{device_id}"
        print(f"Using device: {self.device}")
        self.predictor = kwargs.get("predictor")
        if self.predictor is None:
            warnings.warn(PARITY_NOTE, stacklevel=2)       # said at run time, not only in the docs
            from ..stablenormal import StableNormalPredictorHIP
            from .. import weights as W
            opts = {k: kwargs[k] for k in ("yoso_timestep", "refine_start", "refine_steps", "prediction_type", "processing_resolution", "workspace_bytes") if k in kwargs}
            model_dir = kwargs.get("model_dir")
            if model_dir and os.path.isdir(model_dir) and os.path.isdir(os.path.join(model_dir, "unet")):
                self.predictor = StableNormalPredictorHIP.from_pretrained(model_dir, device_id=device_id, **opts)
            elif kwargs.get("synthetic_weights", False):
                cfgs = kwargs.get("cfgs") or (W.tiny_sn_cfgs() if kwargs.get("tiny", False) else None)
                self.predictor = StableNormalPredictorHIP.from_random(seed=int(kwargs.get("weight_seed", 7)), cfgs=cfgs,
                                                                      device_id=device_id, **opts)
            else:
                raise FileNotFoundError(f"StableNormal checkpoints not found under model_dir={model_dir!r}; pass synthetic_weights=True "
                                        "for seeded random weights of the same architecture, or predictor=<callable>")
        self.parity = getattr(self.predictor, "parity", "caller-supplied predictor")
        print("Model loaded")

    def prepare_input(self, data):
        frames = [np.asarray(x).transpose(1, 2, 0).astype(np.uint8) for x in data["images"]]
        return np.stack(frames, axis=0).astype(np.float32) / 255.0

    @staticmethod
    def postprocess(pred_uint8_list):
        import torch
        ns = [np.array(n) for n in pred_uint8_list]
        for n in ns:
            n[:, :, 0] = -n[:, :, 0]                 # uint8 negate: 0 -> 0, v -> 256 - v
        ns = [n / 255.0 * 2 - 1 for n in ns]
        normals = torch.stack([torch.from_numpy(x).float() for x in ns], dim=0)
        return {"pred_normals": normals, "pred_depths": torch.zeros_like(normals[..., 0])}

    def forward(self, data):
        if hasattr(self.predictor, "predict_batch"):
            from ..stablenormal import normals_to_uint8
            n = self.predictor.predict_batch(self.prepare_input(data))       # uint8-truncated frames, one batch
            out = self.postprocess(list(normals_to_uint8(n)))
            out["parity"] = self.parity                                      # extra key (the harness reads pred_normals / pred_depths only)
            return out
        from PIL import Image
        images = [Image.fromarray(np.asarray(x).transpose(1, 2, 0).astype(np.uint8)) for x in data["images"]]
        return self.postprocess([self.predictor(im) for im in images])
