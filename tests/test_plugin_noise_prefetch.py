"""DepthCrafter.forward draws the NEXT clip's host noise in a background thread while the GPU runs the current one (unigeo_amd/model/depthcrafter.py).
The numbers the pipeline receives must be exactly make_noise(T, H, W, seed + index) whatever the prefetch guessed - serial indices (hits), a sharded
stride (hit from the third call on), a jump (miss), a shape change (miss).  CPU only: the pipeline is a recorder."""
from types import SimpleNamespace

import numpy as np

from unigeo_amd.model.depthcrafter import DepthCrafter
from unigeo_amd.pipeline import make_noise


class _Recorder:
    def __init__(self):
        self.calls = []

    def __call__(self, frames, noise_latents=None, noise_aug=None, seed=None, **kw):
        self.calls.append((frames.shape, seed, noise_latents, noise_aug))
        T, H, W, _ = frames.shape
        return SimpleNamespace(frames=[frames], depth=np.ones((T, H, W), np.float32), normals=np.zeros((T, H, W, 3), np.float32))


def _data(T, H, W, index=None):
    d = {"images": [np.zeros((3, H, W), np.float32) for _ in range(T)], "intrinsics": [np.eye(3, dtype=np.float32)] * T}
    if index is not None:
        d["_index"] = index
    return d


def _plugin(seed=7):
    p = DepthCrafter.__new__(DepthCrafter)
    p.pipeline, p.num_inference_steps, p.seed, p._calls, p.device = _Recorder(), 2, seed, 0, "cpu"
    return p


def test_prefetched_noise_is_the_noise_of_the_seed():
    p = _plugin()
    plan = [(2, 64, 64, 0), (2, 64, 64, 1), (2, 64, 64, 2),          # serial loop: stride 1
            (2, 64, 64, 10), (2, 64, 64, 18), (2, 64, 64, 26),       # a rank of a sharded run: stride 8 (third call on: a hit)
            (2, 64, 64, 3),                                          # a jump back: miss
            (3, 64, 128, 4), (3, 64, 128, 5)]                        # another clip shape: miss, then hit
    for T, H, W, idx in plan:
        out = p.forward(_data(T, H, W, idx))
        assert tuple(out["pred_depths"].shape) == (T, H, W)
    assert len(p.pipeline.calls) == len(plan)
    for (T, H, W, idx), (shape, seed, nl, na) in zip(plan, p.pipeline.calls):
        assert shape == (T, H, W, 3) and seed == 7 + idx
        rl, ra = make_noise(T, H, W, 7 + idx)
        np.testing.assert_array_equal(nl, rl)
        np.testing.assert_array_equal(na, ra)


def test_anonymous_samples_use_the_call_counter():
    p = _plugin(seed=100)
    for k in range(3):
        p.forward(_data(2, 64, 64))
    for k, (shape, seed, nl, na) in enumerate(p.pipeline.calls):
        assert seed == 100 + k
        rl, ra = make_noise(2, 64, 64, 100 + k)
        np.testing.assert_array_equal(nl, rl)
        np.testing.assert_array_equal(na, ra)
