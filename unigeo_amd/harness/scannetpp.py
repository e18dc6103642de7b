"""ScanNet++ clip loader - the sample format either side of the hot path.

Restates, for the processed ScanNet++ layout (``<root>/<scene>/{scene_metadata.npz, images/*.webp, normal/*.webp,
depth/*.png}``):

* sequence metadata + frame sub-sampling + clip split  - ``/root/reference/dataset/scannetpp/scannetpp.py:16-69``
  (OpenCV camera-to-world trajectories -> OpenGL world-to-camera, every 3rd frame, clips every
  ``clip_length - clip_overlap`` frames with the tail padded by its last frame);
* per-clip sample load                                 - ``scannetpp.py:81-135`` (RGB float32 CHW 0..255, normals
  ``x/255*2-1`` with all-zero pixels invalid, depth in mm -> back-projected with the FIRST view's intrinsics,
  flipped to OpenGL);
* post-processing                                      - ``scannetpp.py:140-187`` (world = key-view frame, validity
  mask 1e-3 <= depth <= 80 and finite, extrinsics relative to the key view);
* input / target resizing                              - ``dataset/dataset_core/transforms.py:38-110`` and
  ``dataset/dataset_core/dataset.py:167-170`` (inputs: order-1 with anti-aliasing + intrinsics scaling; targets:
  order-0, no anti-aliasing).

Pinned by ``tests/golden/scannetpp_golden.npz`` (outputs of the reference's own classes on the synthetic scene in
``tests/golden/scannetpp_scene``).  The resize step follows scikit-image's published ``transform.resize`` recipe
(gaussian pre-filter sigma=(s-1)/2 when down-scaling, ``scipy.ndimage.zoom(grid_mode=True, mode='mirror')``) through
scipy; scikit-image itself is not installed here, so that step is UNPINNED (native-size loads are bit-exact vs the reference).
"""
import os

import numpy as np
from PIL import Image

from .dataset import split_clips

_GL_CV = np.float32([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]])
FRAME_GAP = 3                      # scannetpp.py:24-30


class ScannetPPSequence:
    """One scene: poses (world-to-camera, OpenGL), intrinsics, relative file paths and the clip table."""

    def __init__(self, root, scene_name, clip_length=30, clip_overlap=0, gap=FRAME_GAP):
        meta_path = os.path.join(root, scene_name, "scene_metadata.npz")
        if not os.path.isfile(meta_path):
            raise FileNotFoundError(f"ScanNet++ scene metadata missing: {meta_path}")
        with np.load(meta_path) as meta:
            c2w_cv = meta["trajectories"]
            K = meta["intrinsics"]
            names = meta["images"].tolist()
        if not (len(c2w_cv) == len(K) == len(names)):
            raise ValueError(f"{meta_path}: trajectories / intrinsics / images disagree in length")
        w2c_gl = np.linalg.inv(np.einsum("ij,njk,kl->nil", _GL_CV, c2w_cv, _GL_CV))
        self.root, self.scene_name = root, scene_name
        self.extrinsics = w2c_gl[::gap]
        self.intrinsics = K[::gap]
        self.rgb_paths = [os.path.join("images", n + ".webp") for n in names][::gap]
        self.normal_paths = [os.path.join("normal", n + ".webp") for n in names][::gap]
        self.depth_paths = [os.path.join("depth", n + ".png") for n in names][::gap]
        self.clips = split_clips(len(self.rgb_paths), clip_length, clip_overlap)


def _backproject_gl(depth_m, K):
    """utils/geometry_utils.py:246-253 followed by the y/z flip of scannetpp.py:126-127; [3,H,W] float32."""
    h, w = depth_m.shape
    u, v = np.meshgrid(np.arange(w), np.arange(h), indexing="xy")
    x = (u - K[0, 2]) * depth_m / K[0, 0]
    y = (v - K[1, 2]) * depth_m / K[1, 1]
    return np.stack((x, -y, -depth_m), axis=0).astype(np.float32)


def _resize(x, ht, wd, order, anti_alias):
    """scikit-image ``resize`` recipe on the last two axes (mode='reflect' == ndimage 'mirror')."""
    from scipy import ndimage as ndi
    h, w = x.shape[-2:]
    if (h, w) == (ht, wd):
        return x
    lead = x.ndim - 2
    y = x.astype(np.float64) if x.dtype != np.float32 else x
    if anti_alias:
        sig = [0.0] * lead + [max(0.0, (h / ht - 1) / 2), max(0.0, (w / wd - 1) / 2)]
        if any(s > 0 for s in sig):
            y = ndi.gaussian_filter(y, sig, mode="mirror")
    return ndi.zoom(y, [1.0] * lead + [ht / h, wd / w], order=order, mode="mirror", grid_mode=True).astype(x.dtype)


class ScannetPPDataset:
    """Clip-level dataset in the unified sample format (dataset/Readme.md:22-33); ``dataset[i]`` is one clip."""

    base_dataset = "scannetpp"

    def __init__(self, root, scenes=None, split_file=None, split="test", clip_length=17, clip_overlap=0,
                 input_size=None, target_size=None, verbose=False, **_):
        if root is None or not os.path.isdir(root):
            raise FileNotFoundError(f"ScanNet++ root not found: {root!r}")
        if isinstance(scenes, str):
            if scenes != "all":
                raise ValueError('scenes must be a list of scene names, None (the split list) or "all"')
            # explicit opt-in: every processed scene under root, sorted (NOT what the reference evaluates)
            scenes = sorted(d for d in os.listdir(root) if os.path.isfile(os.path.join(root, d, "scene_metadata.npz")))
        elif scenes is None:
            # scannetpp.py:209-217: the scene set and its ORDER come from splits/<split>.txt next to the loader
            # ('train' or, for every other split value, 'nvs_sem_val' - the ten validation scenes, shipped as data)
            if split_file is None:
                split_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "splits",
                                          ("train" if split == "train" else "nvs_sem_val") + ".txt")
            if not os.path.isfile(split_file):
                raise FileNotFoundError(f"ScanNet++ split list not found: {split_file} (pass split_file=... or scenes=[...])")
            with open(split_file) as f:
                scenes = [ln for ln in f.read().splitlines() if ln.strip()]
        self.root, self.split = root, split
        self.input_size, self.target_size = input_size, target_size
        self.samples = []
        for sc in scenes:
            seq = ScannetPPSequence(root, sc, clip_length, clip_overlap)
            if verbose:
                print(f"sequence name: {sc}, num_seq: {len(seq.rgb_paths)}")
            for key, ids in seq.clips.items():
                self.samples.append((seq, key, ids))

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, index):
        if index >= len(self.samples):
            raise IndexError(index)
        seq, _, ids = self.samples[index]
        out = load_clip(self.root, seq, ids)
        out["_index"] = index
        out["_dataset"] = self.base_dataset
        if self.input_size is not None:
            ht, wd = self.input_size
            oh, ow = out["images"][0].shape[-2:]
            out["images"] = [_resize(im, ht, wd, 1, True) for im in out["images"]]
            scale = np.array([[wd / ow] * 3, [ht / oh] * 3, [1.0] * 3], np.float32)
            out["intrinsics"] = [k * scale for k in out["intrinsics"]]
        if self.target_size is not None:
            ht, wd = self.target_size
            for attr in ("cam_normal", "world_normal", "cam_coord", "world_coord", "mask"):
                out[attr] = [_resize(x, ht, wd, 0, False) for x in out[attr]]
        return out


def load_clip(root, seq, ids, keyview_idx=0):
    base = os.path.join(root, seq.scene_name)
    out = {"_base": root, "scene_name": "_".join(seq.scene_name.split("/")), "keyview_idx": keyview_idx, "caption": ""}
    out["images"] = [np.array(Image.open(os.path.join(base, seq.rgb_paths[i]))).astype(np.float32).transpose(2, 0, 1)
                     for i in ids]
    out["image_names"] = [os.path.basename(seq.rgb_paths[i]) for i in ids]
    ext = [seq.extrinsics[i].astype(np.float32) for i in ids]
    out["intrinsics"] = [seq.intrinsics[i].astype(np.float32) for i in ids]
    K0 = out["intrinsics"][0]                                 # scannetpp.py:104 uses view 0's intrinsics for all views

    ref = ext[keyview_idx]
    ref_inv = np.linalg.inv(ref)
    cam_n, cam_c, wor_n, wor_c, masks = [], [], [], [], []
    for j, i in enumerate(ids):
        raw = np.array(Image.open(os.path.join(base, seq.normal_paths[i]))).astype(np.float32)
        hole = np.all(raw < 1e-3, axis=2)
        n = raw / 255.0 * 2 - 1
        n[hole] = 0
        n = n.astype(np.float32).transpose(2, 0, 1)
        depth = np.array(Image.open(os.path.join(base, seq.depth_paths[i]))).astype(np.float32) / 1000
        c = _backproject_gl(depth, K0)
        M = ref @ np.linalg.inv(ext[j])                        # source camera -> key-view camera
        wn = (M[:3, :3] @ n.reshape(3, -1)).reshape(n.shape)
        wc = (M[:3, :3] @ c.reshape(3, -1) + M[:3, 3][:, None]).reshape(c.shape)
        d = -1 * c[2]
        bad = np.isnan(n).any(0) | np.isnan(c).any(0)
        d[np.isnan(d)] = 0
        bad |= (d < 1e-3) | (d > 80)
        for a in (n, c, wn, wc):
            a[:, bad] = 0
        cam_n.append(n); cam_c.append(c); wor_n.append(wn); wor_c.append(wc); masks.append((~bad).astype(np.float32))
    out.update(cam_normal=cam_n, cam_coord=cam_c, world_normal=wor_n, world_coord=wor_c, mask=masks,
               extrinsics=[e @ ref_inv for e in ext])
    return out
