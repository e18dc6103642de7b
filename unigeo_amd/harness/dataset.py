"""Dataset side of the harness.

* ``split_clips`` restates the clip rule of ``/root/reference/dataset/scannetpp/scannetpp.py:41-48``: clips start
  every ``clip_length - clip_overlap`` frames, the last one is padded by repeating its final frame.
* ``SyntheticGeometryDataset`` yields samples in the unified format of ``/root/reference/dataset/Readme.md:22-33``
  (channel-first arrays, OpenGL camera coordinates) from an analytic scene, so the whole harness - model call,
  GT preparation, metrics, CSV - can be exercised without ScanNet++ (not available on the build / bench boxes).
"""
import numpy as np


def split_clips(num_frames, clip_length, clip_overlap):
    clips = {}
    for idx in range(0, num_frames, clip_length - clip_overlap):
        g = list(range(idx, min(idx + clip_length, num_frames)))
        g += [g[-1]] * (clip_length - len(g))
        clips[idx] = g
    return clips


class SyntheticGeometryDataset:
    """A slowly moving camera looking at a wavy surface; GT depth / normals / points are analytic."""

    def __init__(self, root=None, clip_length=25, clip_overlap=5, input_size=(384, 512), target_size=None,
                 num_frames=45, seed=0, **_):
        self.h, self.w = input_size
        self.clips = list(split_clips(num_frames, clip_length, clip_overlap).values())
        self.seed = seed

    def __len__(self):
        return len(self.clips)

    def _frame(self, t):
        H, W = self.h, self.w
        f = 500.0 * (W / 640.0)
        K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], np.float32)
        v, u = np.mgrid[0:H, 0:W].astype(np.float32)
        z = 2.5 + 0.6 * np.sin(u / 40.0 + 0.05 * t) * np.cos(v / 33.0) + 0.002 * v          # OpenCV depth
        x = (u - K[0, 2]) * z / f; y = (v - K[1, 2]) * z / f
        cam_cv = np.stack([x, y, z], 0)
        dzdu = np.gradient(z, axis=1); dzdv = np.gradient(z, axis=0)
        n = np.stack([-dzdu * f / z, -dzdv * f / z, -np.ones_like(z)], 0)
        n /= np.linalg.norm(n, axis=0, keepdims=True)
        gl = np.array([1.0, -1.0, -1.0], np.float32)[:, None, None]
        rng = np.random.default_rng(self.seed * 100003 + t)
        img = np.stack([127.5 + 100 * np.sin(2 * np.pi * (u / 64 + v / 48 + 0.03 * t) + c) for c in range(3)], 0)
        img = np.clip(img + rng.normal(0, 8, img.shape), 0, 255).astype(np.float32)
        ext = np.eye(4, dtype=np.float32); ext[0, 3] = 0.01 * t
        cam_gl = (cam_cv * gl).astype(np.float32)
        world = cam_gl.copy(); world[0] -= 0.01 * t
        return dict(image=img, K=K, ext=ext, cam=cam_gl, normal=(n * gl).astype(np.float32), world=world,
                    mask=(z > 1e-3) & (z < 80))

    def __getitem__(self, i):
        if i >= len(self.clips):
            raise IndexError(i)
        fr = [self._frame(t) for t in self.clips[i]]
        return {"scene_name": "synthetic", "images": [f["image"] for f in fr], "intrinsics": [f["K"] for f in fr],
                "extrinsics": [f["ext"] for f in fr], "cam_coord": [f["cam"] for f in fr],
                "cam_normal": [f["normal"] for f in fr], "world_coord": [f["world"] for f in fr],
                "mask": [f["mask"] for f in fr], "keyview_idx": 0, "_index": i,
                "image_names": [f"frame-{t:06d}.color.png" for t in self.clips[i]]}
