"""Oracle (test infrastructure): depth -> surface normals exactly as the reference wrapper computes them.

Follows /root/reference/model/depthcrafter.py:48-59 (``prepare_output``),
/root/reference/utils/geometry_utils.py:246-253 (``backproject_to_cv_position``: numpy promotes to float64, the
wrapper then casts to float32) and :9-70 (``get_surface_normal``: 5x5 zero-padded box sums of the moments,
per-pixel ``(A^T A + 1e-6 I) n = A^T 1`` by ``torch.linalg.lstsq`` in fp32, L2-normalise, orient towards the camera),
then the y,z negation to OpenGL.  The reference solves in 4x4 image tiles only to bound memory - the per-pixel
systems are independent, so this restatement solves them in one batch.  Pinned against the golden vectors G3
(tests/golden/reference_goldens.npz) by tests/test_reference_goldens.py.
"""
import numpy as np
import torch
import torch.nn.functional as F


def backproject(depth, K):
    h, w = depth.shape
    i, j = np.meshgrid(np.arange(w), np.arange(h), indexing="xy")
    z = depth
    x = (i - K[0, 2]) * z / K[0, 0]
    y = (j - K[1, 2]) * z / K[1, 1]
    return np.stack((x, y, z), axis=-1)


def surface_normal(xyz: torch.Tensor, patch=5) -> torch.Tensor:
    """xyz [H,W,3] fp32 -> unit normals [H,W,3] oriented towards the camera (OpenCV frame)."""
    p = xyz.permute(2, 0, 1)[None]                                   # [1,3,H,W]
    x, y, z = p[:, 0:1], p[:, 1:2], p[:, 2:3]
    k = torch.ones(1, 1, patch, patch)
    box = lambda t: F.conv2d(t, k, padding=patch // 2)[0, 0]
    ata = torch.stack([box(x * x), box(x * y), box(x * z), box(x * y), box(y * y), box(y * z),
                       box(x * z), box(y * z), box(z * z)], -1).reshape(*xyz.shape[:2], 3, 3)
    ata = ata + 1e-6 * torch.eye(3)
    at1 = torch.stack([box(x), box(y), box(z)], -1)[..., None]
    n = torch.linalg.lstsq(ata, at1).solution[..., 0]
    n = n / torch.sqrt((n ** 2).sum(-1, keepdim=True))
    flip = (n * xyz).sum(-1) > 0
    n[flip] *= -1
    return n


def prepare_output(depths, intrinsics):
    """list of [H,W] np depth + list of [3,3] -> (pred_depths [T,H,W], pred_normals [T,H,W,3] OpenGL)."""
    normals = []
    for d, K in zip(depths, intrinsics):
        pts = torch.from_numpy(backproject(d, K)).float()
        n = surface_normal(pts)
        n[:, :, 1:] = -n[:, :, 1:]
        normals.append(n)
    return torch.stack([torch.from_numpy(np.asarray(d)).float() for d in depths], 0), torch.stack(normals, 0)
