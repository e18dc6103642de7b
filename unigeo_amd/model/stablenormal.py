"""StableNormal plugin shell.

Mirrors ``/root/reference/model/stablenormal.py``: ``__init__(**kwargs)`` (:9-18), ``forward(data)`` (:30-52)
with the exact uint8 post-processing (x flip on uint8 wraps mod 256 at :43, ``/255*2-1`` at :45, zero depths).

The predictor itself is ``torch.hub.load("Stable-X/StableNormal", ...)`` (:16): network access plus a
repository whose architecture is not described anywhere in the reference tree, so it cannot be restated
here.  The plugin therefore takes the predictor as an argument (any callable PIL -> PIL/uint8 array) and
raises if none is supplied - there is no silent fallback.  BASELINE config 4 is out of round-1 scope.
"""
import numpy as np


class StableNormal:
    def __init__(self, **kwargs):
        self.predictor = kwargs.get("predictor")
        if self.predictor is None:
            raise NotImplementedError(
                "StableNormal needs the Stable-X/StableNormal hub predictor (network + un-vendored code in the "
                "reference, model/stablenormal.py:16); pass predictor=<callable image -> uint8 normal image>")
        print("Model loaded")

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
        from PIL import Image
        images = [Image.fromarray(np.asarray(x).transpose(1, 2, 0).astype(np.uint8)) for x in data["images"]]
        return self.postprocess([self.predictor(im) for im in images])
