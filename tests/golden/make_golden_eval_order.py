"""Generates tests/golden/eval_order_vectors.npz: what the reference's OWN Scene.__init__ (scene/__init__.py:28-101) makes of the
committed init directory tests/golden/init_scene under `--eval` (build container only: it executes /root/reference).

Under --eval readColmapSceneInfo returns ONE list object as train and test cameras (scene/dataset_readers.py:334-338) and reads
the poses of sparse_<n>/1; Scene.__init__ then shuffles "both" lists — the same list twice — and builds both camera lists from
the twice-shuffled order.  instantsplat_amd.scene_io.load_init_scene(eval=True) must land on the same order (uid = the row of
the pose table, the view-sampling sequence).   Run:  python tests/golden/make_golden_eval_order.py"""
import os
import random
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402
from oracle import knn_ref, raster_torch  # noqa: E402

dgr = types.ModuleType("diff_gaussian_rasterization")
dgr.GaussianRasterizationSettings, dgr.GaussianRasterizer = raster_torch.RasterSettings, object
knn = types.ModuleType("simple_knn._C")
knn.distCUDA2 = lambda pts: knn_ref.dist2(pts)
R = ref_loader.load(lambda m, k, v: m.__setitem__(k, v), lambda m, k: m.pop(k, None), dgr, knn)
_zeros = torch.zeros
torch.zeros = lambda *a, **k: _zeros(*a, **{kk: ("cpu" if (kk == "device" and vv == "cuda") else vv) for kk, vv in k.items()})

out = {}
with tempfile.TemporaryDirectory() as td:
    src = os.path.join(td, "scene")
    shutil.copytree(os.path.join(HERE, "init_scene"), src)
    os.makedirs(os.path.join(src, "sparse_3", "1"))
    for f in ("cameras.txt", "images.txt"):   # the test poses of --eval live in sparse_<n>/1 (here: the same ones)
        shutil.copyfile(os.path.join(src, "sparse_3", "0", f), os.path.join(src, "sparse_3", "1", f))
    args = types.SimpleNamespace(source_path=src, model_path=os.path.join(td, "model"), n_views=3, images=None, eval=True, white_background=False,
                                 resolution=2, data_device="cpu", init_scale_from_view_depth=False, sh_degree=3)
    os.makedirs(args.model_path)
    random.seed(0)
    scene = R.Scene(args, R.gm.GaussianModel(3))
    for kind, cams in (("train", scene.getTrainCameras()), ("test", scene.getTestCameras())):
        out[f"eval_{kind}_names"] = np.array([c.image_name for c in cams])
        out[f"eval_{kind}_uid_colmap"] = np.array([[c.uid, c.colmap_id] for c in cams])
    out["eval_rng_next"] = np.array([random.randint(0, 10 ** 6) for _ in range(4)])
np.savez_compressed(os.path.join(HERE, "eval_order_vectors.npz"), **out)
print({k: v.tolist() for k, v in out.items()})
