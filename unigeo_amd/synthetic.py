"""Synthetic clip in the reference's unified sample format (``/root/reference/dataset/Readme.md:22-33``):
``images`` list of [3,H,W] float32 0..255, ``intrinsics`` list of [3,3] float32 (pixel units).
Recipe from SURVEY.md 8(d): smooth textured video + noise, seeded."""
import numpy as np


def synthetic_clip(T=25, H=384, W=512, seed=1234):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:H, 0:W].astype(np.float32)
    imgs = []
    for t in range(T):
        fr = np.stack([127.5 + 100.0 * np.sin(2 * np.pi * (x / 64 + y / 48 + 0.03 * t) + c) for c in range(3)], 0)
        fr = fr + rng.normal(0, 8, fr.shape)
        imgs.append(np.clip(fr, 0, 255).astype(np.float32))
    f = 500.0 * (W / 640.0)
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], np.float32)
    return {"images": imgs, "intrinsics": [K.copy() for _ in range(T)], "scene_name": f"synthetic_{seed}"}
