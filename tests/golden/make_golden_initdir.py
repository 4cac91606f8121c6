"""Generates tests/golden/init_scene/ (a small init directory in the layout InstantSplat's init stage writes, SURVEY.md
Appendix F) and tests/golden/initdir_vectors.npz (what the reference builds from it, and how it trains on it) by running the
reference's OWN code from /root/reference (build container only; both artefacts are committed):

  the directory      written by the reference's save_extrinsic / save_intrinsics / save_points3D / storePly
                     (utils/sfm_utils.py:202-316,495-510, function definitions executed from the file) + images/*.png via PIL
  initdir_info_*     readColmapSceneInfo + readColmapCameras + getNerfppNorm (scene/dataset_readers.py:50-160,315-369)
  initdir_cam_*      Scene.__init__ (scene/__init__.py:28-101): seeded shuffle, loadCam / PILtoTorch at -r 1 and -r 2
                     (utils/camera_utils.py:21-54), Camera constants, cameras.json, input.ply
  initdir_gm_*       create_from_pcd + init_RT_seq through the real Scene; with --init_scale_from_view_depth as well
  initdir_loop_*     training() (train.py:87-230) with the REAL Scene, prepare_output_and_logger, save_pose and scene.save():
                     view order, losses, final parameters, pose_org / pose_optimized.npy, point_cloud.ply — around the fp32 C
                     oracle as the rasterizer operator (which does not exist here)

Run:  python tests/golden/make_golden_initdir.py"""
import json
import os
import random
import shutil
import sys
import tempfile
import types
from argparse import ArgumentParser, Namespace
from pathlib import Path

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402
from instantsplat_amd.synthetic import syn_pointmap  # noqa: E402  (input data only)
from oracle import gs_ref, knn_ref, raster_torch  # noqa: E402

OUT = os.path.join(HERE, "initdir_vectors.npz")
SCENE_DIR = os.path.join(HERE, "init_scene")


class _OracleRasterizer:
    def __init__(self, raster_settings):
        self.s = raster_settings

    def __call__(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        return gs_ref.rasterize(means3D, means2D, opacities, self.s, shs=shs, colors_precomp=colors_precomp, scales=scales,
                                rotations=rotations, cov3D_precomp=cov3D_precomp)


dgr = types.ModuleType("diff_gaussian_rasterization")
dgr.GaussianRasterizationSettings, dgr.GaussianRasterizer = raster_torch.RasterSettings, _OracleRasterizer
knn = types.ModuleType("simple_knn._C")
knn.distCUDA2 = lambda pts: knn_ref.dist2(pts)
R = ref_loader.load(lambda m, k, v: m.__setitem__(k, v), lambda m, k: m.pop(k, None), dgr, knn)
W = ref_loader.sfm_writers(R.ply)
_zeros = torch.zeros
torch.zeros = lambda *a, **k: _zeros(*a, **{kk: ("cpu" if (kk == "device" and vv == "cuda") else vv) for kk, vv in k.items()})

# ------------------------------------------------------------------------------------------------ the scene on disk
V, WM, IW, IH, ITERS = 3, 6, 64, 48, 10          # images stored at 64x48; training at -r 2 = 32x24 (the emulator's size)
sc = syn_pointmap(V, WM, WM, IW, IH, seed=31)
g = torch.Generator().manual_seed(37)
pts_noisy = (sc.points + 0.01 * torch.randn(sc.points.shape, generator=g)).numpy().astype(np.float32)
col_noisy = (sc.colors + 0.05 * torch.randn(sc.colors.shape, generator=g)).clamp(0, 1).numpy().astype(np.float32)
init_scaling_delta = 0.35 * torch.randn(sc.points.shape[0], 3, generator=g)
init_rotation = torch.randn(sc.points.shape[0], 4, generator=g)
init_rotation = init_rotation / init_rotation.norm(dim=1, keepdim=True) * (0.9 + 0.2 * torch.rand(sc.points.shape[0], 1, generator=g))


class _Pipe:
    compute_cov3D_python = convert_SHs_python = debug = False


def _ref_cam(c, image):
    w2c = c.world_view_transform.t().double().numpy()
    return R.cm.Camera(colmap_id=c.colmap_id, R=w2c[:3, :3].T.copy(), T=w2c[:3, 3].copy(), FoVx=c.FoVx, FoVy=c.FoVy, image=image,
                       gt_alpha_mask=None, image_name=f"v{c.uid}", uid=c.uid, data_device="cpu")


# ground-truth images: the teacher (create_from_pcd of the clean points) rendered at the true poses by the oracle, 64x48
teacher = R.gm.GaussianModel(3)
teacher.create_from_pcd(R.graphics_utils.BasicPointCloud(points=sc.points.numpy(), colors=sc.colors.numpy(),
                                                         normals=np.zeros((sc.points.shape[0], 3))), sc.extent)
blank = torch.zeros(3, IH, IW)
teacher.init_RT_seq({1.0: [_ref_cam(c, blank) for c in sc.cameras]})
with torch.no_grad():
    gts = [R.gr.render(_ref_cam(c, blank), teacher, _Pipe, torch.zeros(3), camera_pose=teacher.get_RT(c.uid))["render"].clamp(0, 1)
           for c in sc.cameras]
# the estimated (noisy) poses an InstantSplat initialisation hands over, as world-to-camera matrices
w2c_est = []
for v, c in enumerate(sc.cameras):
    p = R.pose_utils.get_tensor_from_camera(c.world_view_transform.t())
    p = torch.cat([R.pose_utils.quadmultiply(sc.pose_noise_q[v:v + 1], p[None, :4])[0], p[4:] + sc.pose_noise_t[v]])
    w2c_est.append(R.pose_utils.get_camera_from_tensor(p).double().numpy())
NAMES = ["view_b.png", "view_c.png", "view_a.png"]   # COLMAP ids 1, 2, 3 in THIS order: sorting by name is not sorting by id

shutil.rmtree(SCENE_DIR, ignore_errors=True)
sparse0 = Path(SCENE_DIR) / f"sparse_{V}" / "0"
sparse0.mkdir(parents=True)
(Path(SCENE_DIR) / "images").mkdir()
focal_mast3r = IW / 2 / (2 * np.tan(sc.cameras[0].FoVx / 2))      # the init stage works on a half-size frame (here 32 px wide) ...
W.save_extrinsic(sparse0, w2c_est, NAMES, ".png")
W.save_intrinsics(sparse0, [focal_mast3r] * V, (IW, IH), (V, IH // 2, IW // 2, 3), save_focals=True)   # ... and rescales the focal
with tempfile.TemporaryDirectory() as td:
    W.save_points3D(sparse0, col_noisy.reshape(V, -1, 3), pts_noisy.reshape(V, -1, 3), sc.confidence.numpy().reshape(V, -1),
                    masks=None, use_masks=False, save_all_pts=False, save_txt_path=td)
from PIL import Image  # noqa: E402
for name, img in zip(NAMES, gts):
    Image.fromarray((img.permute(1, 2, 0).numpy() * 255.0).round().astype(np.uint8)).save(os.path.join(SCENE_DIR, "images", name))
for junk in ("images.bin", "cameras.bin", "confidence.npy", "non_scaled_focals.npy"):   # written by the init stage, never read by train
    os.remove(sparse0 / junk)
print("wrote", SCENE_DIR, sorted(os.listdir(sparse0)))

out = {"initdir_config": np.array([V, WM, IW, IH, ITERS], dtype=np.int64), "initdir_init_scaling_delta": init_scaling_delta.numpy(),
       "initdir_init_rotation": init_rotation.numpy()}

# ------------------------------------------------------------------------------------------------ readColmapSceneInfo
args = types.SimpleNamespace(n_views=V)
info = R.dr.readColmapSceneInfo(SCENE_DIR, None, False, args)
out["initdir_info_names"] = np.array([c.image_name for c in info.train_cameras])
out["initdir_info_uid"] = np.array([c.uid for c in info.train_cameras])
out["initdir_info_R"], out["initdir_info_T"] = np.stack([c.R for c in info.train_cameras]), np.stack([c.T for c in info.train_cameras])
out["initdir_info_fov"] = np.array([[c.FovX, c.FovY] for c in info.train_cameras], dtype=np.float64)
out["initdir_info_wh"] = np.array([[c.width, c.height] for c in info.train_cameras])
out["initdir_info_radius"], out["initdir_info_translate"] = np.float64(info.nerf_normalization["radius"]), info.nerf_normalization["translate"]
out["initdir_info_points"], out["initdir_info_colors"] = np.asarray(info.point_cloud.points), np.asarray(info.point_cloud.colors)
out["initdir_info_poses"] = np.stack(info.train_poses)
assert len(info.test_cameras) == 0

# ------------------------------------------------------------------------------------------------ Scene.__init__
def _scene_args(model_path, resolution, view_depth=False):
    return types.SimpleNamespace(source_path=SCENE_DIR, model_path=model_path, n_views=V, images=None, eval=False, white_background=False,
                                 resolution=resolution, data_device="cpu", init_scale_from_view_depth=view_depth, sh_degree=3)


for res in (1, 2):
    with tempfile.TemporaryDirectory() as td:
        random.seed(0)
        gmod = R.gm.GaussianModel(3)
        scene = R.Scene(_scene_args(td, res), gmod)
        cams = scene.getTrainCameras()
        p = f"initdir_cam_r{res}_"
        out[p + "names"] = np.array([c.image_name for c in cams])
        out[p + "uid_colmap"] = np.array([[c.uid, c.colmap_id] for c in cams])
        out[p + "wh"] = np.array([[c.image_width, c.image_height] for c in cams])
        out[p + "fov"] = np.array([[c.FoVx, c.FoVy] for c in cams], dtype=np.float64)
        out[p + "world_view_transform"] = np.stack([c.world_view_transform.numpy() for c in cams])
        out[p + "projection_matrix"] = np.stack([c.projection_matrix.numpy() for c in cams])
        out[p + "camera_center"] = np.stack([c.camera_center.numpy() for c in cams])
        out[p + "original_image"] = np.stack([c.original_image.numpy() for c in cams])
        if res == 2:
            out["initdir_cameras_extent"] = np.float64(scene.cameras_extent)
            out["initdir_cameras_json"] = np.array(open(os.path.join(td, "cameras.json")).read())
            assert open(os.path.join(td, "input.ply"), "rb").read() == open(sparse0 / "points3D.ply", "rb").read()
            for n in ("_xyz", "_features_dc", "_scaling", "_rotation", "_opacity", "P"):
                out["initdir_gm" + (n if n.startswith("_") else "_" + n)] = getattr(gmod, n).detach().numpy().copy()
            out["initdir_rng_next"] = np.array([random.randint(0, 10 ** 6) for _ in range(4)])   # where the `random` stream is after the shuffle
with tempfile.TemporaryDirectory() as td:
    random.seed(0)
    gmod = R.gm.GaussianModel(3)
    R.Scene(_scene_args(td, 2, view_depth=True), gmod)
    out["initdir_gm_scaling_view_depth"] = gmod._scaling.detach().numpy().copy()

# ------------------------------------------------------------------------------------------------ training() on the directory
track = {"models": [], "uids": [], "l1": [], "loss": []}


class _TrackedModel(R.gm.GaussianModel):
    def __init__(self, sh_degree):
        super().__init__(sh_degree)
        track["models"].append(self)


class _SceneFromDisk(R.Scene):
    """The reference's Scene, unchanged, followed by the same generic start the other loop goldens use: create_from_pcd leaves
    every Gaussian isotropic, where d(loss)/d(rotation) is rounding noise that Adam's first steps turn into +-lr moves — two
    correct implementations separate there for reasons that have nothing to do with the loop."""

    def __init__(self, args, gaussians, *a, **k):
        super().__init__(args, gaussians, *a, **k)
        with torch.no_grad():
            gaussians._scaling.add_(init_scaling_delta)
            gaussians._rotation.copy_(init_rotation)


class _Quiet:
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return lambda *a, **k: 0.0


def _render_tracked(cam, *a, **k):
    track["uids"].append(cam.uid)
    return R.gr.render(cam, *a, **k)


def _l1_tracked(a, b):
    v = R.loss_utils.l1_loss(a, b)
    track["l1"].append(v.detach())
    return v


def _ssim_tracked(a, b):
    v = R.loss_utils.ssim(a, b)
    track["loss"].append(float((1.0 - 0.2) * track["l1"][-1] + 0.2 * (1.0 - v.detach())))
    return v


fns = R.train_functions
tns = {"os": os, "np": np, "torch": torch, "Namespace": Namespace, "TENSORBOARD_FOUND": False, "GaussianModel": _TrackedModel,
       "Scene": _SceneFromDisk, "tqdm": _Quiet, "time": __import__("time").time, "randint": random.randint, "render": _render_tracked,
       "l1_loss": _l1_tracked, "ssim": _ssim_tracked, "FUSED_SSIM_AVAILABLE": False, "save_time": lambda *a, **k: None,
       "training_report": lambda *a, **k: None, "_Quiet": _Quiet, "get_camera_from_tensor": R.pose_utils.get_camera_from_tensor}
for name in ("load_and_prepare_confidence", "save_pose", "prepare_output_and_logger", "training"):
    code = ref_loader.cpu(fns[name]).replace("torch.cuda.Event(enable_timing = True)", "_Quiet()")
    exec(compile(code, os.path.join(REF, "train.py"), "exec"), tns)
opt = R.OptimizationParams(ArgumentParser())
opt.iterations, opt.pp_optimizer, opt.optim_pose = ITERS, True, True
with tempfile.TemporaryDirectory() as td:
    dataset = _scene_args(os.path.join(td, "model"), 2)
    random.seed(0)
    tns["training"](dataset, opt, _Pipe, [], [ITERS], [], None, -1)
    model = track["models"][-1]
    mp = dataset.model_path
    out["initdir_loop_view_uids"], out["initdir_loop_losses"] = np.array(track["uids"]), np.array(track["loss"], dtype=np.float64)
    for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P"):
        out["initdir_loop_final" + (n if n.startswith("_") else "_" + n)] = getattr(model, n).detach().numpy().copy()
    out["initdir_loop_final_steps"] = np.array([model.optimizer.state[grp["params"][0]]["step"] for grp in model.optimizer.param_groups])
    out["initdir_loop_pose_org"] = np.load(os.path.join(mp, "pose", f"ours_{ITERS}", "pose_org.npy"))
    out["initdir_loop_pose_optimized"] = np.load(os.path.join(mp, "pose", f"ours_{ITERS}", "pose_optimized.npy"))
    ply = R.ply.PlyData.read(os.path.join(mp, "point_cloud", f"iteration_{ITERS}", "point_cloud.ply")).elements[0]
    out["initdir_loop_ply_names"] = np.array(list(ply.data.dtype.names))
    out["initdir_loop_ply_columns"] = np.stack([ply.data[n] for n in ply.data.dtype.names], axis=1)
    out["initdir_loop_outputs"] = np.array(sorted(os.path.relpath(os.path.join(d, f), mp) for d, _, fs in os.walk(mp) for f in fs))
    assert "cfg_args" in out["initdir_loop_outputs"] and "input.ply" in out["initdir_loop_outputs"]
torch.zeros = _zeros
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays")
print("views", out["initdir_loop_view_uids"], "losses", np.round(out["initdir_loop_losses"], 5))
print("names after sort", out["initdir_info_names"], "after shuffle", out["initdir_cam_r2_names"], out["initdir_cam_r2_uid_colmap"].tolist())
print("extent", out["initdir_cameras_extent"], "outputs", out["initdir_loop_outputs"])
