#!/usr/bin/env python
"""Golden for the ScanNet++ sample loader: writes a tiny synthetic scene in the processed
ScanNet++ layout the reference reads (scene_metadata.npz + images/*.webp + normal/*.webp + depth/*.png, see
/root/reference/dataset/scannetpp/scannetpp.py:50-69,81-135) under tests/golden/scannetpp_scene/, runs the
REFERENCE's own ScannetPPSequence (gap=3, clip split :25-48) and ScannetPPSample.load/postprocess (:81-187) on it
and stores what they return.  Third-party modules the reference imports but this path never calls are stubbed.
Runs only in the build container."""
import importlib.abc
import importlib.machinery
import os
import shutil
import sys
import types

import numpy as np
from PIL import Image

OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(OUT, "scannetpp_scene")
SCENE = "sceneA"


class _Dummy:
    def __call__(self, *a, **k): return _Dummy()
    def __getattr__(self, k): return _Dummy()
    def __mro_entries__(self, bases): return (object,)


class _Any(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Dummy()


PREF = ("cv2", "open3d", "torchvision", "h5py", "skimage", "pytoml", "roma", "imageio", "matplotlib", "tensorboard",
        "wandb", "torch.utils.tensorboard", "evo", "trimesh", "pyrender")


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.startswith(PREF):
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
    def create_module(self, spec):
        m = _Any(spec.name); m.__path__ = []; return m
    def exec_module(self, m): pass


def write_scene():
    shutil.rmtree(ROOT, ignore_errors=True)
    d = os.path.join(ROOT, SCENE)
    for sub in ("images", "normal", "depth"):
        os.makedirs(os.path.join(d, sub))
    rng = np.random.default_rng(7)
    n, H, W = 10, 24, 32
    names, poses, Ks = [], [], []
    for i in range(n):
        nm = f"frame_{i:06d}"
        names.append(nm)
        Image.fromarray(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).save(os.path.join(d, "images", nm + ".webp"), lossless=True)
        nrm = rng.integers(1, 256, (H, W, 3), dtype=np.uint8); nrm[:2, :3] = 0            # all-zero pixels = invalid normal
        Image.fromarray(nrm).save(os.path.join(d, "normal", nm + ".webp"), lossless=True)
        dep = rng.integers(300, 6000, (H, W)).astype(np.uint16); dep[5, 5] = 0; dep[6, 6] = 65000   # 0 m: masked; 65 m: valid (uint16 mm cannot exceed 80 m)
        Image.fromarray(dep).save(os.path.join(d, "depth", nm + ".png"))
        a = 0.05 * i
        P = np.eye(4, dtype=np.float64); P[:3, :3] = [[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]
        P[:3, 3] = [0.1 * i, 0.02 * i, -0.05 * i]
        poses.append(P)
        Ks.append(np.array([[40.0, 0, W / 2], [0, 40.0, H / 2], [0, 0, 1]], np.float64))
    np.savez(os.path.join(d, "scene_metadata.npz"), trajectories=np.stack(poses), intrinsics=np.stack(Ks), images=np.array(names))


def main():
    write_scene()
    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, "/root/reference")
    import dataset.scannetpp.scannetpp as sp
    seq = sp.ScannetPPSequence(ROOT, SCENE, clip_length=3, clip_overlap=1)
    G = {"clip_keys": np.array(list(seq.source_ids.keys())), "clip_ids": np.array(list(seq.source_ids.values()))}
    for ci, (key, ids) in enumerate(seq.source_ids.items()):
        s = sp.ScannetPPSample(base=ROOT, name=SCENE)
        s.data = {"images": [seq.rgb_path_list[i] for i in ids], "poses": [seq.extrinsics[i] for i in ids],
                  "intrinsics": [seq.intrinsics[i] for i in ids], "depth": [seq.depth_path_list[i] for i in ids],
                  "keyview_idx": 0, "normal": [seq.normal_path_list[i] for i in ids]}
        out = s.load(ROOT)
        for k in ("images", "extrinsics", "intrinsics", "cam_normal", "cam_coord", "world_normal", "world_coord", "mask"):
            G[f"c{ci}_{k}"] = np.stack([np.asarray(x) for x in out[k]])
        G[f"c{ci}_names"] = np.array(out["image_names"])
        G[f"c{ci}_scene"] = np.array(out["scene_name"])
    np.savez_compressed(os.path.join(OUT, "scannetpp_golden.npz"), **G)
    print({k: v.shape for k, v in G.items()})


if __name__ == "__main__":
    main()
