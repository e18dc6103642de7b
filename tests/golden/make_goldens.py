#!/usr/bin/env python
"""Generate golden vectors by running the REFERENCE's own Python (imported from /root/reference, read-only)
on seeded synthetic inputs.  Runs only in the build container; the .npz/.csv fixtures it writes travel,
the reference does not.  Modules the reference imports but the hot path never calls are stubbed.

Covers SURVEY.md 8(c) G1-G7: wrapper prepare_input / post-processing / prepare_output (back-projection +
surface normals + OpenGL flip), prepare_gt_label, depth_evaluation (lstsq, custom mask), normal_evaluation,
MetricsManager CSV, StableNormal uint8 post-processing.
"""
import importlib.util
import io
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

for name in ["cv2", "open3d", "torchvision", "h5py", "skimage", "pytoml", "roma", "imageio", "evo", "evo.core",
             "evo.core.trajectory", "evo.tools", "evo.core.metrics", "evo.core.sync", "evo.tools.file_interface",
             "evo.core.geometry", "evo.tools.plot", "matplotlib", "matplotlib.pyplot"]:
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
sys.path.insert(0, REF)


def load_by_path(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


geom = load_by_path("utils.geometry_utils", "utils/geometry_utils.py")
sys.modules.setdefault("utils", types.ModuleType("utils")).__path__ = [os.path.join(REF, "utils")]
sys.modules["utils.geometry_utils"] = geom
io_utils = load_by_path("utils.io_utils", "utils/io_utils.py")
dc = load_by_path("ref_depthcrafter", "model/depthcrafter.py")          # bypasses the broken model/__init__.py:5
align = load_by_path("metrics.alignment", "metrics/alignment.py")
sys.modules.setdefault("metrics", types.ModuleType("metrics")).__path__ = [os.path.join(REF, "metrics")]
sys.modules["metrics.alignment"] = align
ev_depth = load_by_path("metrics.eval_depth", "metrics/eval_depth.py")
ev_normal = load_by_path("metrics.eval_normal", "metrics/eval_normal.py")
save_utils = load_by_path("metrics.save_utils", "metrics/save_utils.py")

rng = np.random.default_rng(20250704)
G = {}

# G1 prepare_input: non-integer float images pin the uint8 truncation (model/depthcrafter.py:39-45)
imgs = [rng.uniform(0, 255.99, (3, 16, 24)).astype(np.float32) for _ in range(3)]
wrapper = dc.DepthCrafter.__new__(dc.DepthCrafter)
G["g1_images"] = np.stack(imgs)
G["g1_frames"] = wrapper.prepare_input({"images": imgs})

# G2 wrapper post-processing, the three lines at model/depthcrafter.py:92-97 executed verbatim
res = rng.uniform(0, 1, (3, 16, 24, 3)).astype(np.float32)
G["g2_res"] = res.copy()
r = res.sum(-1) / res.shape[-1]
r = (r - r.min()) / (r.max() - r.min())
G["g2_depths"] = np.stack([1 / (x + 0.1) for x in r])

# G3 prepare_output: back-projection + get_surface_normal (+ y,z flip) on a smooth synthetic depth
H, W, T = 48, 64, 2
yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
depths = [(2.0 + 0.8 * np.sin(xx / 9.0 + 0.5 * t) * np.cos(yy / 7.0) + 0.02 * yy + 0.01 * xx).astype(np.float32)
          for t in range(T)]
K = np.array([[60.0, 0, W / 2.0], [0, 60.0, H / 2.0], [0, 0, 1]], np.float32)
torch.manual_seed(0)
out = wrapper.prepare_output([d.copy() for d in depths], {"intrinsics": [K] * T})
G["g3_depths"] = np.stack(depths); G["g3_K"] = np.stack([K] * T)
G["g3_pred_depths"] = out["pred_depths"].numpy(); G["g3_pred_normals"] = out["pred_normals"].numpy()

# G4 prepare_gt_label (utils/io_utils.py:4-45)
Tn, h, w = 2, 8, 12
data = {"images": [rng.uniform(0, 255, (3, h, w)).astype(np.float32) for _ in range(Tn)],
        "extrinsics": [np.eye(4, dtype=np.float32) + 0.01 * rng.standard_normal((4, 4)).astype(np.float32) for _ in range(Tn)],
        "world_coord": [rng.standard_normal((3, h, w)).astype(np.float32) for _ in range(Tn)],
        "cam_coord": [rng.standard_normal((3, h, w)).astype(np.float32) for _ in range(Tn)],
        "mask": [rng.uniform(size=(h, w)) > 0.3 for _ in range(Tn)],
        "cam_normal": [rng.standard_normal((3, h, w)).astype(np.float32) for _ in range(Tn)]}
for k, v in data.items():
    G["g4_in_" + k] = np.stack(v)
gt = io_utils.prepare_gt_label({k: [a.copy() for a in v] for k, v in data.items()})
for k, v in gt.items():
    G["g4_out_" + k] = v.numpy()

# G5 depth_evaluation(align_with_lstsq=True, custom_mask) and normal_evaluation (eval.py:47-56)
Nf, h, w = 3, 20, 28
gt_d = rng.uniform(0.5, 6.0, (Nf, h, w)).astype(np.float32)
gt_d[0, :3] = 0.0; gt_d[1, 5, 5] = 90.0            # invalid pixels: <=0 and >= max_depth
pred_d = (0.7 * gt_d + 0.3 + 0.2 * rng.standard_normal(gt_d.shape)).astype(np.float32)
mask = rng.uniform(size=gt_d.shape) > 0.2
dres = ev_depth.depth_evaluation(torch.from_numpy(pred_d), torch.from_numpy(gt_d), custom_mask=torch.from_numpy(mask),
                                 align_with_lstsq=True)[0]
G["g5_gt_d"], G["g5_pred_d"], G["g5_mask"] = gt_d, pred_d, mask
G["g5_depth_keys"] = np.array(list(dres.keys())); G["g5_depth_vals"] = np.array([float(v) for v in dres.values()], np.float64)
gn = rng.standard_normal((Nf, h, w, 3)).astype(np.float32); gn /= np.linalg.norm(gn, axis=-1, keepdims=True)
pn = gn + 0.3 * rng.standard_normal(gn.shape).astype(np.float32)
nres = ev_normal.normal_evaluation(torch.from_numpy(pn), torch.from_numpy(gn), custom_mask=torch.from_numpy(mask))
G["g5_gt_n"], G["g5_pred_n"] = gn, pn
G["g5_normal_keys"] = np.array(list(nres.keys())); G["g5_normal_vals"] = np.array([float(v) for v in nres.values()], np.float64)

# G6 MetricsManager CSV text (metrics/save_utils.py:5-90)
names = ["Abs Rel", "delta < 1.25", "normal mean", "angle < 11.25"]
mm = save_utils.MetricsManager(metric_names=names)
rows = [{"seq_name": "000_sceneA", "Abs Rel": 0.123456789, "delta < 1.25": 0.87654321, "normal mean": 21.5, "angle < 11.25": 33.333333},
        {"seq_name": "001_sceneB", "Abs Rel": 0.2, "delta < 1.25": 0.75, "normal mean": 19.25}]
csv_path = os.path.join(OUT, "_tmp", "metrics.csv")
for r_ in rows:
    mm.update_metrics(dict(r_))
    mm.export_to_csv(csv_path)
open(os.path.join(OUT, "metrics_manager.csv"), "w").write(open(csv_path).read())
os.remove(csv_path); os.rmdir(os.path.dirname(csv_path))

# G7 StableNormal post-processing (model/stablenormal.py:40-51): uint8 negate wraps mod 256
pred = [rng.integers(0, 256, (6, 8, 3), dtype=np.uint8) for _ in range(2)]
pred[0][0, 0, 0] = 0; pred[0][0, 1, 0] = 255; pred[0][0, 2, 0] = 128
G["g7_in"] = np.stack(pred)
pn_ = [np.array(n) for n in pred]
for i in range(len(pn_)):
    pn_[i][:, :, 0] = -pn_[i][:, :, 0]
pn_ = [n / 255. * 2 - 1 for n in pn_]
G["g7_normals"] = torch.stack([torch.from_numpy(x).float() for x in pn_], dim=0).numpy()

np.savez_compressed(os.path.join(OUT, "reference_goldens.npz"), **G)
print("wrote", os.path.join(OUT, "reference_goldens.npz"), {k: v.shape for k, v in G.items()})

# G3 at full frame size ("48x64 and one 384x512 frame"; round 6): the reference's own prepare_output on ONE 384x512 frame - the depth and intrinsics
# of tests/test_pipeline_gpu.py::test_normals_kernel_vs_reference_golden_full_frame.  Its own file (2 MB), so that the fixtures above keep their bytes.
H, W = 384, 512
yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
d_full = (3.0 + np.sin(xx / 50.0) * np.cos(yy / 37.0) + 0.003 * yy).astype(np.float32)
f_ = 500.0 * (W / 640.0)
K_full = np.array([[f_, 0, W / 2.0], [0, f_, H / 2.0], [0, 0, 1]], np.float32)          # = unigeo_amd.synthetic.synthetic_clip's intrinsics
torch.manual_seed(0)
out = wrapper.prepare_output([d_full.copy()], {"intrinsics": [K_full]})
np.savez_compressed(os.path.join(OUT, "reference_g3_fullframe.npz"), g3f_depth=d_full, g3f_K=K_full,
                    g3f_pred_depths=out["pred_depths"].numpy(), g3f_pred_normals=out["pred_normals"].numpy())
print("wrote reference_g3_fullframe.npz", out["pred_normals"].shape)
