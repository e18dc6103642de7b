"""DepthCrafter plugin on the MI355X-native engine.

Mirrors ``/root/reference/model/depthcrafter.py`` member for member:
  ``__init__(model_dir, unet_path, pre_train_path, **kwargs)`` (:8-36), ``prepare_input`` (:39-45),
  ``prepare_output`` (:48-69), ``forward(data) -> {'pred_depths' [Nf,H,W], 'pred_normals' [Nf,H,W,3]}``
  (:73-99), both CPU float32 torch tensors, normals in OpenGL camera coordinates.

Differences that are deliberate and visible:
  * the denoise loop, VAE, CLIP, the depth post-processing AND the back-projection / surface-normal
    step (40 s per clip on the reference's CPU path) run on the GPU inside one C-ABI call;
  * noise comes from a seeded CPU generator (``seed`` kwarg + the sample's dataset index, so clips get independent noise
    like the reference's fresh draws) instead of the un-seeded global CUDA RNG;
  * ``num_inference_steps`` defaults to the reference's shipped value 5 (:86) and is a kwarg;
  * without checkpoints on disk the constructor raises unless ``synthetic_weights=True`` is passed.
"""
import os

import numpy as np

from ..pipeline import DepthCrafterPipelineHIP


class DepthCrafter:
    def __init__(self, model_dir=None, unet_path=None, pre_train_path=None, **kwargs):
        self.num_inference_steps = int(kwargs.get("num_inference_steps", 5))
        self.seed = int(kwargs.get("seed", 0))
        self._calls = 0
        device_id = int(kwargs.get("device_id", 0))
        self.device = f"hip:{device_id}"
        print(f"Using device: {self.device}")
        have = bool(unet_path) and bool(pre_train_path) and os.path.isdir(unet_path) and os.path.isdir(pre_train_path)
        if have:
            kw = {"workspace_bytes": kwargs["workspace_bytes"]} if kwargs.get("workspace_bytes") else {}
            self.pipeline = DepthCrafterPipelineHIP.from_pretrained(pre_train_path, unet_path, device_id=device_id, **kw)
        elif kwargs.get("synthetic_weights", False):
            cfgs = kwargs.get("cfgs")
            if cfgs is None and kwargs.get("tiny", False):      # YAML-selectable tiny full-topology configuration (plumbing checks)
                from .. import weights as W
                cfgs = W.tiny_cfgs()
            self.pipeline = DepthCrafterPipelineHIP.from_random(seed=int(kwargs.get("weight_seed", 42)),
                                                                cfgs=cfgs, device_id=device_id,
                                                                workspace_bytes=kwargs.get("workspace_bytes"))
        else:
            raise FileNotFoundError(f"checkpoints not found (unet_path={unet_path!r}, pre_train_path={pre_train_path!r}); "
                                    "pass synthetic_weights=True for seeded random weights of the same architecture")
        self.pipeline.to(self.device)
        self.pipeline.enable_xformers_memory_efficient_attention()   # no-ops kept for call compatibility
        self.pipeline.enable_attention_slicing()
        print(f"Model loaded from {unet_path}")

    def prepare_input(self, data):
        frames = [np.asarray(x).transpose(1, 2, 0).astype(np.uint8) for x in data["images"]]
        return np.stack(frames, axis=0).astype(np.float32) / 255.0

    def prepare_output(self, depthcrafter_depths, data, _device_normals=None):
        """Reference signature (model/depthcrafter.py:48): list / array of [H,W] depth maps + the sample dict (``intrinsics``) ->
        {'pred_depths' [Nf,H,W], 'pred_normals' [Nf,H,W,3]} in OpenGL camera coordinates.  The back-projection + surface-normal
        fit + y/z flip of :51-59 run on the GPU (``k_normals``); ``forward`` passes the normals ``ug_dc_run`` already produced."""
        import torch
        depths = np.ascontiguousarray(np.stack([np.asarray(d, dtype=np.float32) for d in depthcrafter_depths], 0))
        if _device_normals is None:
            K = np.stack([np.asarray(k, dtype=np.float32).reshape(3, 3) for k in data["intrinsics"]], 0)
            _device_normals = self.pipeline.engine.normals_from_depth(depths, K)
        return {"pred_depths": torch.from_numpy(depths).float(),
                "pred_normals": torch.from_numpy(np.ascontiguousarray(_device_normals)).float()}

    # ---- noise prefetch (host side; see forward)
    def _noise_for(self, shape, seed):
        from ..pipeline import make_noise
        pf = getattr(self, "_noise_pf", None)
        if pf is not None:
            self._noise_pf = None
            key, thread, box = pf
            thread.join()
            if key == (shape, seed) and "noise" in box:
                return box["noise"]
        return make_noise(shape[0], shape[1], shape[2], seed)

    def _noise_prefetch(self, shape, seed):
        import threading
        from ..pipeline import make_noise
        box = {}

        def work():
            try:
                box["noise"] = make_noise(shape[0], shape[1], shape[2], seed)
            except Exception:      # a failed guess is not an error: forward draws the noise itself
                pass
        th = threading.Thread(target=work, daemon=True)
        th.start()
        self._noise_pf = ((shape, seed), th, box)

    def forward(self, data):
        frames = self.prepare_input(data)
        K = np.stack([np.asarray(k, dtype=np.float32).reshape(3, 3) for k in data["intrinsics"]], 0)
        # the reference draws fresh noise per clip from the global RNG; here: a per-clip seed derived from the sample's
        # dataset index (reproducible, rank-independent in sharded runs), or a call counter for anonymous samples
        clip_seed = self.seed + int(data["_index"]) if "_index" in data else self.seed + self._calls
        self._calls += 1
        # The host noise draw (torch CPU generator, ~30 ms per 25 x 384 x 512 clip) is taken off the critical path: the noise of the clip this instance will most
        # likely see next (same shape; seed advanced by the stride of the last two seeds - 1 in the serial loop, the world size in sharded runs) is drawn by a
        # background thread while the GPU runs this clip.  Same seeds, same generator, same numbers: a wrong guess only wastes the draw.
        shape = (int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2]))
        noise_latents, noise_aug = self._noise_for(shape, clip_seed)
        last = getattr(self, "_last_seed", None)
        stride = clip_seed - last if last is not None and clip_seed != last else 1
        self._last_seed = clip_seed
        self._noise_prefetch(shape, clip_seed + stride)
        res = self.pipeline(frames, height=frames.shape[1], width=frames.shape[2], output_type="np",
                            guidance_scale=1.0, num_inference_steps=self.num_inference_steps,
                            window_size=len(frames), overlap=25, track_time=False, seed=clip_seed,
                            noise_latents=noise_latents, noise_aug=noise_aug,
                            intrinsics=K, with_normals=True, return_frames=False)
        return self.prepare_output(list(res.depth), data, _device_normals=res.normals)
