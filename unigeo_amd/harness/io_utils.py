"""GT preparation - mirrors ``/root/reference/utils/io_utils.py:4-45`` (``prepare_gt_label``): the unified sample
dict (OpenGL camera coordinates, channel-first arrays) -> stacked torch tensors in OpenCV coordinates."""
import numpy as np

_GL2CV = np.diag([1.0, -1.0, -1.0, 1.0]).astype(np.float32)


def prepare_gt_label(data):
    import torch
    n = len(data["images"])
    world, masks, poses, depths, rgbs, normals = [], [], [], [], [], []
    for i in range(n):
        cam2world = np.linalg.inv(np.asarray(data["extrinsics"][i]).astype(np.float32))
        poses.append(np.einsum("ij,jk,kl->il", _GL2CV, cam2world, _GL2CV))
        pw = np.asarray(data["world_coord"][i]).astype(np.float32).copy(); pw[1:] *= -1      # y,z flip
        pc = np.asarray(data["cam_coord"][i]).astype(np.float32).copy(); pc[1:] *= -1
        world.append(pw.transpose(1, 2, 0)); depths.append(pc.transpose(1, 2, 0)[..., -1])
        masks.append(np.asarray(data["mask"][i]).astype(bool))
        rgbs.append(np.asarray(data["images"][i]).transpose(1, 2, 0) / 255.0)
        normals.append(np.asarray(data["cam_normal"][i]).transpose(1, 2, 0))
    st = lambda xs: torch.from_numpy(np.ascontiguousarray(np.stack(xs, 0)))
    return {"gt_world_pts": st(world), "gt_masks": st(masks), "gt_poses": st(poses), "gt_depths": st(depths),
            "gt_rgbs": st(rgbs), "gt_normals": st(normals)}
