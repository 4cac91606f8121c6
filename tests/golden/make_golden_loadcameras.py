"""tests/golden/loadcameras_vectors.npz: the reference's own `loadCameras` (scene/dataset_readers.py:75-104) and the
SIMPLE_PINHOLE branch of `readColmapCameras` (:129-132), run from /root/reference (build container only).
Run:  python tests/golden/make_golden_loadcameras.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT); sys.path.insert(0, REF); sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

dgr, knn = types.ModuleType("diff_gaussian_rasterization"), types.ModuleType("simple_knn._C")
dgr.GaussianRasterizationSettings = dgr.GaussianRasterizer = object
knn.distCUDA2 = lambda p: None
R = ref_loader.load(lambda m, k, v: m.__setitem__(k, v), lambda m, k: m.pop(k, None), dgr, knn)
g = torch.Generator().manual_seed(5)


def rand_w2c():
    q = torch.randn(4, generator=g).double()
    q = (q / q.norm()).numpy()
    m = np.eye(4)
    m[:3, :3] = R.cl.qvec2rotmat(q)
    m[:3, 3] = torch.randn(3, generator=g).double().numpy()
    return m


def cam(i):
    m = rand_w2c()
    return R.cm.Camera(colmap_id=i + 1, R=m[:3, :3].T.copy(), T=m[:3, 3].copy(), FoVx=0.9, FoVy=0.7, image=torch.zeros(3, 6, 8), gt_alpha_mask=None,
                       image_name=f"v{i}", uid=i, data_device="cpu")


out = {}
for tag, n in (("same", 3), ("longer", 7)):
    cams = [cam(i) for i in range(3)]
    poses = np.stack([rand_w2c() for _ in range(n)])
    out[f"lc_{tag}_in_R"] = np.stack([c.R for c in cams]); out[f"lc_{tag}_in_T"] = np.stack([c.T for c in cams])
    out[f"lc_{tag}_poses"] = poses
    res = R.dr.loadCameras(poses, cams)
    out[f"lc_{tag}_uid_colmap"] = np.array([[c.uid, c.colmap_id] for c in res]); out[f"lc_{tag}_names"] = np.array([c.image_name for c in res])
    out[f"lc_{tag}_R"], out[f"lc_{tag}_T"] = np.stack([c.R for c in res]), np.stack([c.T for c in res])
    for k in ("world_view_transform", "full_proj_transform", "camera_center"):
        out[f"lc_{tag}_{k}"] = np.stack([getattr(c, k).numpy() for c in res])
# SIMPLE_PINHOLE: readColmapCameras' FoV for one focal
Intr = R.cl.Camera
Extr = R.cl.Image if hasattr(R.cl, "Image") else R.cl.BaseImage
from PIL import Image  # noqa: E402
import tempfile  # noqa: E402
with tempfile.TemporaryDirectory() as td:
    Image.fromarray(np.zeros((6, 8, 3), dtype=np.uint8)).save(os.path.join(td, "a.png"))
    intr = {1: Intr(id=1, model="SIMPLE_PINHOLE", width=8, height=6, params=np.array([7.5, 4.0, 3.0]))}
    extr = {1: Extr(id=1, qvec=np.array([1.0, 0.0, 0.0, 0.0]), tvec=np.array([0.1, 0.2, 3.0]), camera_id=1, name="a.png", xys=np.zeros((0, 2)), point3D_ids=np.zeros(0))}
    infos, _ = R.dr.readColmapCameras(extr, intr, td)
out["simple_pinhole_fov"] = np.array([infos[0].FovX, infos[0].FovY])
np.savez_compressed(os.path.join(HERE, "loadcameras_vectors.npz"), **out)
print("wrote", len(out), "arrays", out["lc_longer_names"], out["simple_pinhole_fov"])
