"""Generates tests/golden/reference_vectors.npz by IMPORTING the reference's own pure-PyTorch helpers
from /root/reference (possible only in the build container; the GPU box has no /root/reference, so the
vectors are committed).  These pin the oracle / product restatements of every hot-path piece that
exists in the reference tree as Python:

  ssim(), l1_loss()            reference utils/loss_utils.py:39-85   (== fused_ssim's definition)
  eval_sh(), RGB2SH            reference utils/sh_utils.py:57-117
  build_covariance...          reference utils/general_utils.py:64-110 (+ scene/gaussian_model.py:32-36)
  getProjectionMatrix          reference utils/graphics_utils.py:71-91
  get_expon_lr_func            reference utils/general_utils.py:29-62
  get_camera_from_tensor, quadmultiply, get_tensor_from_camera   reference utils/pose_utils.py:10-215
  PerPointAdam.step            reference scene/per_point_adam.py:34-100
  psnr                         reference utils/image_utils.py:17-19
  load_and_prepare_confidence  reference train.py:63-85 (function definition executed on its own)
  GaussianModel                reference scene/gaussian_model.py:29-243 (create_from_pcd, init_RT_seq, activations,
                               training_setup_pp, update_learning_rate, oneupSHdegree; module executed from its file)
  Camera                       reference scene/cameras.py:17-57 (per-view constants from the reference's own class)
  training()                   reference train.py:87-230 — the loop itself, executed around the fp32 C oracle as the operator:
                               per-iteration losses, view order, LR schedule, optimizer steps, final parameters
  save_pose()                  reference train.py:46-60
  render_set_optimize()        reference render.py:99-186 — test-view pose tracking, executed the same way: pose sequence,
                               masked-L1 losses, best pose, final rendering
  render()                     reference gaussian_renderer/__init__.py:23-144 — the arguments it passes to the rasterizer
                               operator (recorded with a stand-in operator), default pipeline and both python-flag variants;
                               and the gradients autograd carries back through that glue from a linear stand-in operator

Run:  python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.npz")

from utils import general_utils, graphics_utils, image_utils, loss_utils, pose_utils, sh_utils  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_ppa", os.path.join(REF, "scene", "per_point_adam.py"))
ppa = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ppa)

g = torch.Generator().manual_seed(20260923)
rn = lambda *s: torch.randn(*s, generator=g)
ru = lambda *s: torch.rand(*s, generator=g)
out = {}

# ---- SSIM / L1 (value and gradient w.r.t. img1)
for name, (H, W) in {"a": (40, 36), "b": (17, 53)}.items():
    x = ru(1, 3, H, W).requires_grad_(True)
    y = (x.detach() + 0.1 * rn(1, 3, H, W)).clamp(0, 1)
    v = loss_utils.ssim(x, y)
    v.backward()
    out[f"ssim_{name}_x"], out[f"ssim_{name}_y"] = x.detach().numpy(), y.numpy()
    out[f"ssim_{name}_val"], out[f"ssim_{name}_grad"] = v.detach().numpy(), x.grad.numpy().copy()
    x.grad = None
    l1 = loss_utils.l1_loss(x, y)
    l1.backward()
    out[f"l1_{name}_val"], out[f"l1_{name}_grad"] = l1.detach().numpy(), x.grad.numpy().copy()

# ---- SH
sh = rn(50, 3, 16)
d = rn(50, 3)
d = d / d.norm(dim=1, keepdim=True)
out["sh_coeffs"], out["sh_dirs"] = sh.numpy(), d.numpy()
for deg in range(4):
    out[f"sh_eval_deg{deg}"] = sh_utils.eval_sh(deg, sh, d).numpy()
out["rgb2sh"] = sh_utils.RGB2SH(torch.tensor([0.0, 0.25, 1.0])).numpy()

# ---- covariance from scaling / rotation (the reference hard-codes device="cuda": strip it for the CPU)
_zeros = torch.zeros
torch.zeros = lambda *a, **k: _zeros(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
try:
    s = torch.exp(0.3 * rn(40, 3))
    q = rn(40, 4)
    L = general_utils.build_scaling_rotation(1.7 * s, q)
    cov = general_utils.strip_symmetric(L @ L.transpose(1, 2))
finally:
    torch.zeros = _zeros
out["cov_scales"], out["cov_rots"], out["cov_mod"], out["cov_packed"] = s.numpy(), q.numpy(), np.float32(1.7), cov.numpy()

# ---- projection
out["proj_args"] = np.array([0.01, 100.0, 1.1, 0.7])
out["proj_matrix"] = graphics_utils.getProjectionMatrix(0.01, 100.0, 1.1, 0.7).numpy()

# ---- LR schedule
f = general_utils.get_expon_lr_func(lr_init=1.6e-4 * 3.0, lr_final=1.6e-6 * 3.0, lr_delay_mult=0.01, max_steps=30000)
steps = np.array([0, 1, 10, 500, 1000, 29999, 30000, 40000])
out["lr_steps"], out["lr_values"] = steps, np.array([f(int(t)) for t in steps])

# ---- pose algebra
pose = torch.cat([rn(4), rn(3)])
out["pose7"] = pose.numpy()
w2c = pose_utils.get_camera_from_tensor(pose)
out["pose_w2c"] = w2c.numpy()
out["pose_back"] = pose_utils.get_tensor_from_camera(w2c).numpy()
q1, q2 = rn(4), rn(30, 4)
out["qm_q1"], out["qm_q2"], out["qm_out"] = q1.numpy(), q2.numpy(), pose_utils.quadmultiply(q1, q2).numpy()

# ---- PerPointAdam trajectory (xyz-like tensor with per-point multiplier, plus a plain tensor, plus a zero-grad step)
p1 = rn(25, 3).requires_grad_(True)
p2 = rn(25, 1, 3).requires_grad_(True)
pplr = 1.0 + 99.0 * ru(25, 1)
opt = ppa.PerPointAdam([{"params": [p1], "per_point_lr": pplr, "lr": 1.6e-4, "name": "xyz"},
                        {"params": [p2], "lr": 2.5e-2, "name": "f_dc"}], lr=0, betas=(0.9, 0.999), eps=1e-15, weight_decay=0.0)
out["adam_p1_0"], out["adam_p2_0"], out["adam_pplr"] = p1.detach().numpy().copy(), p2.detach().numpy().copy(), pplr.numpy()
grads1, grads2 = [], []
for t in range(4):
    g1 = rn(25, 3) if t != 2 else torch.zeros(25, 3)  # step 2: all-zero gradient -> moments frozen, update still applied
    g2 = rn(25, 1, 3)
    p1.grad, p2.grad = g1.clone(), g2.clone()
    opt.step()
    grads1.append(g1.numpy())
    grads2.append(g2.numpy())
    out[f"adam_p1_{t + 1}"], out[f"adam_p2_{t + 1}"] = p1.detach().numpy().copy(), p2.detach().numpy().copy()
out["adam_g1"], out["adam_g2"] = np.stack(grads1), np.stack(grads2)

# ---- psnr
a, b = ru(3, 8, 9), ru(3, 8, 9)
out["psnr_a"], out["psnr_b"], out["psnr_val"] = a.numpy(), b.numpy(), image_utils.psnr(a, b).numpy()

np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays")

# ---- COLMAP text readers (reference scene/colmap_loader.py, loaded by path: scene/__init__ needs plyfile)
spec2 = importlib.util.spec_from_file_location("ref_colmap", os.path.join(REF, "scene", "colmap_loader.py"))
cl = importlib.util.module_from_spec(spec2)
spec2.loader.exec_module(cl)
HERE = os.path.dirname(os.path.abspath(__file__))
cams = cl.read_intrinsics_text(os.path.join(HERE, "colmap_cameras.txt"))
imgs = cl.read_extrinsics_text(os.path.join(HERE, "colmap_images.txt"))
ids = sorted(cams)
out2 = dict(np.load(OUT))
out2["colmap_cam_ids"] = np.array(ids)
out2["colmap_cam_wh"] = np.array([[cams[i].width, cams[i].height] for i in ids])
out2["colmap_cam_params"] = np.stack([cams[i].params for i in ids])
iid = sorted(imgs)
out2["colmap_img_ids"] = np.array(iid)
out2["colmap_img_qvec"] = np.stack([imgs[i].qvec for i in iid])
out2["colmap_img_tvec"] = np.stack([imgs[i].tvec for i in iid])
out2["colmap_img_camid"] = np.array([imgs[i].camera_id for i in iid])
out2["colmap_img_R"] = np.stack([cl.qvec2rotmat(imgs[i].qvec) for i in iid])
np.savez_compressed(OUT, **out2)
print("added COLMAP vectors:", len(out2), "arrays")

# ---- MASt3R confidence -> per-point LR multipliers (reference train.py:63-85; train.py itself cannot be imported here —
# torchvision / the CUDA operators are missing — so only that function's definition is taken from the file and executed)
import ast  # noqa: E402
import tempfile  # noqa: E402

src = open(os.path.join(REF, "train.py")).read()
fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "load_and_prepare_confidence")
ns = {"np": np, "torch": torch}
exec(compile(ast.Module(body=[fn], type_ignores=[]), os.path.join(REF, "train.py"), "exec"), ns)
g3 = torch.Generator().manual_seed(7)
conf = (3.0 + 2.0 * torch.randn(64, 1, generator=g3)).numpy().astype(np.float32)
with tempfile.TemporaryDirectory() as td:
    np.save(os.path.join(td, "confidence_dsp.npy"), conf)
    lr_mod = ns["load_and_prepare_confidence"](os.path.join(td, "confidence_dsp.npy"), device="cpu", scale=(1, 100))
out3 = dict(np.load(OUT))
out3["confidence_raw"], out3["confidence_lr_modifiers"] = conf, lr_mod.numpy()
np.savez_compressed(OUT, **out3)
print("added confidence vectors:", len(out3), "arrays")

# ---- GaussianModel (reference scene/gaussian_model.py:29-243): initialisation from a point cloud, parameter layouts,
# activations, optimiser groups and LR schedule, produced by the reference's OWN class.  The module is executed from its
# file with (a) `.cuda()` / device="cuda" rewritten to the CPU in memory, (b) stand-ins for the two imports that do not
# exist here: `plyfile` (unused by the methods called) and `simple_knn._C.distCUDA2` (the exact 3-NN mean squared distance,
# computed by oracle/knn_ref.py's float64 k-d tree), (c) `scene` registered as a bare namespace package so that
# `scene.per_point_adam` loads without scene/__init__.py (which needs plyfile / PIL readers).
import types  # noqa: E402
from argparse import ArgumentParser  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import knn_ref  # noqa: E402

knn_mod = types.ModuleType("simple_knn._C")
knn_mod.distCUDA2 = lambda pts: knn_ref.dist2(pts)
sys.modules["simple_knn"] = types.ModuleType("simple_knn")
sys.modules["simple_knn._C"] = knn_mod
ply_mod = types.ModuleType("plyfile")
ply_mod.PlyData = ply_mod.PlyElement = object
sys.modules["plyfile"] = ply_mod
scene_pkg = types.ModuleType("scene")
scene_pkg.__path__ = [os.path.join(REF, "scene")]
sys.modules["scene"] = scene_pkg
gm_path = os.path.join(REF, "scene", "gaussian_model.py")
gm_src = open(gm_path).read().replace(".cuda()", "").replace('device="cuda"', 'device="cpu"')
gm = types.ModuleType("ref_gaussian_model")
exec(compile(gm_src, gm_path, "exec"), gm.__dict__)
from arguments import OptimizationParams as RefOptimizationParams  # noqa: E402
from utils.graphics_utils import BasicPointCloud, getWorld2View2  # noqa: E402

g4 = torch.Generator().manual_seed(11)
n_pts = 96
pts = (torch.rand(n_pts, 3, generator=g4) * 2 - 1).numpy().astype(np.float32)
cols = torch.rand(n_pts, 3, generator=g4).numpy().astype(np.float32)
conf_lr = (1.0 + 99.0 * torch.rand(n_pts, 1, generator=g4)).float()


class _Cam:  # the one attribute init_RT_seq reads (reference scene/cameras.py:52)
    def __init__(self, R, t):
        self.world_view_transform = torch.tensor(getWorld2View2(R, t, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)


cam_R, cam_t = [], []
for k in range(3):
    a = 0.2 * (k - 1)
    cam_R.append(np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], dtype=np.float64))
    cam_t.append(np.array([0.1 * k, -0.05 * k, 4.0 + 0.3 * k], dtype=np.float64))
ref_model = gm.GaussianModel(3)
ref_model.create_from_pcd(BasicPointCloud(points=pts, colors=cols, normals=np.zeros_like(pts)), 2.5)
ref_model.init_RT_seq({1.0: [_Cam(R, t) for R, t in zip(cam_R, cam_t)]})
ropt = RefOptimizationParams(ArgumentParser())
ropt.iterations = 1000
ref_model.training_setup_pp(ropt, conf_lr)
out4 = dict(np.load(OUT))
out4["gm_points"], out4["gm_colors"], out4["gm_conf_lr"] = pts, cols, conf_lr.numpy()
out4["gm_cam_R"], out4["gm_cam_t"] = np.stack(cam_R), np.stack(cam_t)
for name in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "P"):
    out4["gm" + name if name.startswith("_") else "gm_" + name] = getattr(ref_model, name).detach().numpy().copy()
out4["gm_get_scaling"], out4["gm_get_opacity"] = ref_model.get_scaling.detach().numpy(), ref_model.get_opacity.detach().numpy()
out4["gm_get_features"] = ref_model.get_features.detach().numpy()
out4["gm_get_rotation"] = ref_model.get_rotation.detach().numpy()
torch.zeros = lambda *a, **k: _zeros(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})  # general_utils: device="cuda"
try:
    out4["gm_get_covariance"] = ref_model.get_covariance(1.3).detach().numpy()
finally:
    torch.zeros = _zeros
out4["gm_group_names"] = np.array([grp["name"] for grp in ref_model.optimizer.param_groups])
lr_rows = []
for it in (1, 2, 10, 500, 1000):
    ref_model.update_learning_rate(it)
    lr_rows.append([grp["lr"] for grp in ref_model.optimizer.param_groups])
out4["gm_lr_iterations"], out4["gm_group_lrs"] = np.array([1, 2, 10, 500, 1000]), np.array(lr_rows, dtype=np.float64)
ref_model.oneupSHdegree()
out4["gm_sh_degree_after_oneup"] = np.array(ref_model.active_sh_degree)
np.savez_compressed(OUT, **out4)
print("added GaussianModel vectors:", len(out4), "arrays")

# ---- render() glue (reference gaussian_renderer/__init__.py:23-144): what the reference hands to the rasterizer operator.
# The operator itself does not exist here, so a recording stand-in is registered under its module name and the reference's
# own render() is executed (device strings rewritten as above) for the default pipeline and both python-flag variants; the
# recorded keyword arguments and settings are the golden values for instantsplat_amd.gaussian_renderer.render().
import collections  # noqa: E402

_FIELDS = ["image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix", "sh_degree",
           "campos", "prefiltered", "debug"]
_rec = {}


class _RecordingRasterizer:
    def __init__(self, raster_settings):
        self.s = raster_settings

    def __call__(self, **kw):
        _rec.clear()
        _rec.update(kw)
        _rec["settings"] = self.s
        return torch.zeros(3, self.s.image_height, self.s.image_width), torch.ones(kw["means3D"].shape[0], dtype=torch.int32)


dgr_mod = types.ModuleType("diff_gaussian_rasterization")
dgr_mod.GaussianRasterizationSettings = collections.namedtuple("GaussianRasterizationSettings", _FIELDS)
dgr_mod.GaussianRasterizer = _RecordingRasterizer
sys.modules["diff_gaussian_rasterization"] = dgr_mod
gr_path = os.path.join(REF, "gaussian_renderer", "__init__.py")
gr_src = open(gr_path).read().replace(".cuda()", "").replace('device="cuda"', 'device="cpu"')
gr = types.ModuleType("ref_gaussian_renderer")
exec(compile(gr_src, gr_path, "exec"), gr.__dict__)

g5 = torch.Generator().manual_seed(13)
with torch.no_grad():
    ref_model._rotation.copy_(torch.randn(n_pts, 4, generator=g5) * (0.8 + 0.4 * torch.rand(n_pts, 1, generator=g5)))
    ref_model._scaling.copy_(torch.log(torch.tensor(0.05)) + 0.5 * torch.randn(n_pts, 3, generator=g5))
    ref_model._opacity.copy_(1.5 * torch.randn(n_pts, 1, generator=g5))
    ref_model._features_dc.copy_(0.5 * torch.randn(n_pts, 1, 3, generator=g5))
    ref_model._features_rest.copy_(0.1 * torch.randn(n_pts, 15, 3, generator=g5))
    ref_model._xyz.add_(torch.tensor([0.0, 0.0, 4.0]))
ref_model.active_sh_degree = 2
pose7 = torch.tensor([0.9, 0.05, -0.1, 0.02, 0.1, -0.2, 0.3])   # un-normalised quaternion (w, x, y, z) + translation


class _ViewCam:
    FoVx, FoVy, image_height, image_width = 1.0, 0.8, 48, 64
    projection_matrix = graphics_utils.getProjectionMatrix(0.01, 100.0, 1.0, 0.8).transpose(0, 1)
    camera_center = torch.tensor([0.3, -0.2, 0.1])


class _Pipe:
    def __init__(self, cov, sh):
        self.compute_cov3D_python, self.convert_SHs_python, self.debug = cov, sh, False


out5 = dict(np.load(OUT))
for k in ("_xyz", "_rotation", "_scaling", "_opacity", "_features_dc", "_features_rest"):
    out5["render_in" + k] = getattr(ref_model, k).detach().numpy().copy()
out5["render_in_pose"], out5["render_in_bg"] = pose7.numpy(), np.array([0.1, 0.2, 0.3], dtype=np.float32)
out5["render_in_camera_center"] = _ViewCam.camera_center.numpy()
torch.zeros = lambda *a, **k: _zeros(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})  # general_utils: device="cuda"
try:
    for tag, (cov, shp) in {"default": (False, False), "cov_python": (True, False), "sh_python": (False, True)}.items():
        gr.render(_ViewCam, ref_model, _Pipe(cov, shp), torch.tensor([0.1, 0.2, 0.3]), scaling_modifier=1.2, camera_pose=pose7)
        for k in ("means3D", "means2D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp"):
            v = _rec[k]
            out5[f"render_{tag}_{k}"] = np.zeros(0, dtype=np.float32) if v is None else v.detach().numpy().copy()
        s_ = _rec["settings"]
        out5[f"render_{tag}_settings_scalars"] = np.array([s_.image_height, s_.image_width, s_.tanfovx, s_.tanfovy, s_.scale_modifier,
                                                           s_.sh_degree, float(s_.prefiltered), float(s_.debug)], dtype=np.float64)
        out5[f"render_{tag}_viewmatrix"], out5[f"render_{tag}_projmatrix"] = s_.viewmatrix.numpy(), s_.projmatrix.numpy()
        out5[f"render_{tag}_campos"], out5[f"render_{tag}_bg"] = s_.campos.numpy(), s_.bg.numpy()
finally:
    torch.zeros = _zeros
np.savez_compressed(OUT, **out5)
print("added render() glue vectors:", len(out5), "arrays")

# ---- Camera (reference scene/cameras.py:17-57): per-view constants, from the reference's own class (device rewritten)
cam_path = os.path.join(REF, "scene", "cameras.py")
cm = types.ModuleType("ref_cameras")
exec(compile(open(cam_path).read().replace(".cuda()", ""), cam_path, "exec"), cm.__dict__)
g6 = torch.Generator().manual_seed(17)
img = torch.rand(3, 20, 28, generator=g6) * 1.4 - 0.2          # exercises the clamp to [0, 1]
a6 = 0.35
R6 = np.array([[np.cos(a6), 0.0, np.sin(a6)], [0.0, 1.0, 0.0], [-np.sin(a6), 0.0, np.cos(a6)]]) @ \
    np.array([[1.0, 0.0, 0.0], [0.0, np.cos(0.2), -np.sin(0.2)], [0.0, np.sin(0.2), np.cos(0.2)]])
T6 = np.array([0.3, -0.4, 3.5])
rc = cm.Camera(colmap_id=5, R=R6, T=T6, FoVx=1.05, FoVy=0.8, image=img, gt_alpha_mask=None, image_name="v", uid=2, data_device="cpu")
out6 = dict(np.load(OUT))
out6["camera_R"], out6["camera_T"], out6["camera_fov"], out6["camera_image_in"] = R6, T6, np.array([1.05, 0.8]), img.numpy()
out6["camera_world_view_transform"], out6["camera_projection_matrix"] = rc.world_view_transform.numpy(), rc.projection_matrix.numpy()
out6["camera_full_proj_transform"], out6["camera_center"] = rc.full_proj_transform.numpy(), rc.camera_center.numpy()
out6["camera_original_image"] = rc.original_image.numpy()
out6["camera_scalars"] = np.array([rc.image_width, rc.image_height, rc.znear, rc.zfar, rc.uid, rc.colmap_id], dtype=np.float64)
np.savez_compressed(OUT, **out6)
print("added Camera vectors:", len(out6), "arrays")

# ---- gradients through the render() glue: the stand-in operator now returns an image that is LINEAR in the tensors it is
# handed, with fixed random coefficients, so d(image.sum())/d(operator input) is known and autograd carries it through the
# reference's own pose transform / activations back to the raw parameters and the 7-vector camera pose.
g7 = torch.Generator().manual_seed(19)
_coef = {k: torch.randn(*shape, generator=g7) for k, shape in
         {"means3D": (n_pts, 3), "rotations": (n_pts, 4), "scales": (n_pts, 3), "opacities": (n_pts, 1)}.items()}


class _LinearRasterizer(_RecordingRasterizer):
    def __call__(self, **kw):
        tot = sum((kw[k] * c).sum() for k, c in _coef.items())
        return tot.expand(3, self.s.image_height, self.s.image_width) / (3 * self.s.image_height * self.s.image_width), \
            torch.ones(kw["means3D"].shape[0], dtype=torch.int32)


gr.GaussianRasterizer = _LinearRasterizer
pose_leaf = pose7.clone().requires_grad_(True)
for p_ in (ref_model._xyz, ref_model._rotation, ref_model._scaling, ref_model._opacity):
    p_.grad = None
res = gr.render(_ViewCam, ref_model, _Pipe(False, False), torch.tensor([0.1, 0.2, 0.3]), scaling_modifier=1.0, camera_pose=pose_leaf)
res["render"].sum().backward()
out7 = dict(np.load(OUT))
for k, c in _coef.items():
    out7["glue_coef_" + k] = c.numpy()
for name in ("_xyz", "_rotation", "_scaling", "_opacity"):
    out7["glue_grad" + name] = getattr(ref_model, name).grad.numpy().copy()
out7["glue_grad_pose"] = pose_leaf.grad.numpy().copy()
np.savez_compressed(OUT, **out7)
print("added glue-gradient vectors:", len(out7), "arrays")

# ---- the training loop itself (reference train.py:87-230, `training()`): LR schedule, view sampling, render, loss,
# backward, PerPointAdam step, the skipped optimizer step of the last iteration — driven by the reference's OWN function.
# Only the function definition is taken from train.py (the file imports torchvision / tensorboard / the CUDA operators) and
# executed with: the reference GaussianModel / render() / Camera / l1_loss / ssim / PerPointAdam / load_and_prepare_confidence
# loaded above, a Scene stand-in that does what reference scene/__init__.py:85-101 does with an in-memory point cloud
# (create_from_pcd + init_RT_seq), no-op logging, and — as the rasterizer operator, which does not exist — the fp32 C oracle
# (oracle/gs_ref.py, operator calling convention).  The trajectory therefore pins the LOOP (everything around the operator).
import random  # noqa: E402
from instantsplat_amd.synthetic import syn_pointmap  # noqa: E402  (input data only: cameras, points, colours, confidence, pose noise)
from oracle import gs_ref, raster_torch  # noqa: E402


class _OracleRasterizer:
    def __init__(self, raster_settings):
        self.s = raster_settings

    def __call__(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        return gs_ref.rasterize(means3D, means2D, opacities, self.s, shs=shs, colors_precomp=colors_precomp, scales=scales,
                                rotations=rotations, cov3D_precomp=cov3D_precomp)


gr.GaussianRasterizationSettings = raster_torch.RasterSettings
gr.GaussianRasterizer = _OracleRasterizer
TL_V, TL_WM, TL_W, TL_H, TL_ITERS = 3, 6, 32, 24, 12
sc8 = syn_pointmap(TL_V, TL_WM, TL_WM, TL_W, TL_H, seed=21)
g8 = torch.Generator().manual_seed(23)
pts_noisy = (sc8.points + 0.01 * torch.randn(sc8.points.shape, generator=g8)).numpy()
col_noisy = (sc8.colors + 0.05 * torch.randn(sc8.colors.shape, generator=g8)).clamp(0, 1).numpy()
bg8 = torch.zeros(3)
init_scaling_delta = 0.35 * torch.randn(sc8.points.shape[0], 3, generator=g8)
init_rotation = torch.randn(sc8.points.shape[0], 4, generator=g8)
init_rotation = init_rotation / init_rotation.norm(dim=1, keepdim=True) * (0.9 + 0.2 * torch.rand(sc8.points.shape[0], 1, generator=g8))


def _ref_cam(c, image):
    w2c = c.world_view_transform.t().double().numpy()
    return cm.Camera(colmap_id=c.colmap_id, R=w2c[:3, :3].T.copy(), T=w2c[:3, 3].copy(), FoVx=c.FoVx, FoVy=c.FoVy, image=image,
                     gt_alpha_mask=None, image_name=f"v{c.uid}", uid=c.uid, data_device="cpu")


teacher = gm.GaussianModel(3)
teacher.create_from_pcd(BasicPointCloud(points=sc8.points.numpy(), colors=sc8.colors.numpy(), normals=np.zeros((sc8.points.shape[0], 3))),
                        sc8.extent)
blank = torch.zeros(3, TL_H, TL_W)
teacher.init_RT_seq({1.0: [_ref_cam(c, blank) for c in sc8.cameras]})
with torch.no_grad():
    gts8 = [gr.render(_ref_cam(c, blank), teacher, _Pipe(False, False), bg8, camera_pose=teacher.get_RT(c.uid))["render"].clamp(0, 1)
            for c in sc8.cameras]
tl = {"models": [], "uids": [], "l1": [], "loss": [], "params": [], "grads": []}
TL_NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P")


class _TrackedModel(gm.GaussianModel):
    def __init__(self, sh_degree):
        super().__init__(sh_degree)
        tl["models"].append(self)


class _Scene:
    def __init__(self, args, gaussians, *a, **k):
        self.model_path, self.cameras_extent = args.model_path, sc8.extent
        self.train_cameras = {1.0: [_ref_cam(c, gts8[c.uid]) for c in sc8.cameras]}
        gaussians.create_from_pcd(BasicPointCloud(points=pts_noisy, colors=col_noisy, normals=np.zeros_like(pts_noisy)), self.cameras_extent, None)
        gaussians.init_RT_seq(self.train_cameras)
        with torch.no_grad():   # the estimated (noisy) camera poses an InstantSplat initialisation would hand over
            P = gaussians.P.detach().clone()
            P[:, :4] = pose_utils.quadmultiply(sc8.pose_noise_q, P[:, :4])
            P[:, 4:] += sc8.pose_noise_t
        gaussians.P = P.requires_grad_(True)
        with torch.no_grad():
            # create_from_pcd leaves every Gaussian isotropic with identity rotation, where d(loss)/d(rotation) is exactly
            # zero and any two correct implementations hold different rounding noise there — which Adam's first steps turn
            # into +-lr moves, so trajectories would separate for reasons that have nothing to do with the loop.  Start the
            # student from anisotropic scales and generic rotations instead (a state any later iteration is in anyway).
            gaussians._scaling.add_(init_scaling_delta)
            gaussians._rotation.copy_(init_rotation)

    def getTrainCameras(self, scale=1.0):
        return self.train_cameras[scale]

    def save(self, iteration):
        pass


class _Bar:
    def __init__(self, *a, **k):
        pass

    def set_postfix(self, *a, **k):
        pass

    def update(self, *a, **k):
        pass

    def close(self):
        pass


class _Ev:
    """Stand-in for the two torch.cuda.Event objects of train.py:120-121.  The reference records `iter_start` as the first
    statement of an iteration (train.py:140) and `iter_end` right after `loss.backward()` (train.py:178), which makes them
    the two points at which a teacher-forced comparison needs the state: parameters going into the iteration, and the
    gradients its backward produced (before the optimizer consumes them)."""

    def record(self):
        if not tl["models"]:
            return
        m = tl["models"][-1]
        if all(getattr(m, n).grad is None for n in TL_NAMES):
            tl["params"].append({n: getattr(m, n).detach().clone() for n in TL_NAMES})
        else:
            tl["grads"].append({n: (torch.zeros_like(getattr(m, n)) if getattr(m, n).grad is None else getattr(m, n).grad.detach().clone())
                                for n in TL_NAMES})

    def elapsed_time(self, other):
        return 0.0


def _render_tracked(cam, *a, **k):
    tl["uids"].append(cam.uid)
    return gr.render(cam, *a, **k)


def _l1_tracked(a, b):
    v = loss_utils.l1_loss(a, b)
    tl["l1"].append(v.detach())
    return v


def _ssim_tracked(a, b):
    v = loss_utils.ssim(a, b)
    tl["loss"].append(float((1.0 - 0.2) * tl["l1"][-1] + 0.2 * (1.0 - v.detach())))
    return v


tsrc = open(os.path.join(REF, "train.py")).read()
tfn = next(n for n in ast.parse(tsrc).body if isinstance(n, ast.FunctionDef) and n.name == "training")
tcode = ast.get_source_segment(tsrc, tfn).replace('device="cuda"', 'device="cpu"').replace("device='cuda'", "device='cpu'") \
    .replace(".cuda()", "").replace("torch.cuda.Event(enable_timing = True)", "_Ev()")
tns = {"os": os, "np": np, "torch": torch, "prepare_output_and_logger": lambda d: None, "GaussianModel": _TrackedModel,
       "load_and_prepare_confidence": ns["load_and_prepare_confidence"], "Scene": _Scene, "save_pose": lambda *a, **k: None,
       "tqdm": _Bar, "time": __import__("time").time, "randint": random.randint, "render": _render_tracked,
       "l1_loss": _l1_tracked, "ssim": _ssim_tracked, "FUSED_SSIM_AVAILABLE": False, "save_time": lambda *a, **k: None,
       "training_report": lambda *a, **k: None, "_Ev": _Ev}
exec(compile(tcode, os.path.join(REF, "train.py"), "exec"), tns)
def _run_reference_training(prefix, pp_optimizer, optim_pose, iters):
    for k in ("models", "uids", "l1", "loss", "params", "grads"):
        tl[k].clear()
    topt = RefOptimizationParams(ArgumentParser())
    topt.iterations, topt.pp_optimizer, topt.optim_pose = iters, pp_optimizer, optim_pose
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, f"sparse_{TL_V}", "0"))
        np.save(os.path.join(td, f"sparse_{TL_V}", "0", "confidence_dsp.npy"), sc8.confidence.numpy())
        dataset = types.SimpleNamespace(sh_degree=3, source_path=td, model_path=td, n_views=TL_V, white_background=False)
        random.seed(0)
        tns["training"](dataset, topt, _Pipe(False, False), [], [], [], None, -1)
    model = tl["models"][-1]
    o = dict(np.load(OUT))
    o[prefix + "_view_uids"], o[prefix + "_losses"] = np.array(tl["uids"]), np.array(tl["loss"], dtype=np.float64)
    for name in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P"):
        o[prefix + "_final" + (name if name.startswith("_") else "_" + name)] = getattr(model, name).detach().numpy().copy()
    o[prefix + "_final_lrs"] = np.array([grp["lr"] for grp in model.optimizer.param_groups], dtype=np.float64)
    o[prefix + "_final_steps"] = np.array([model.optimizer.state[grp["params"][0]].get("step", 0) if grp["params"][0] in model.optimizer.state
                                            else 0 for grp in model.optimizer.param_groups])
    o[prefix + "_flags"] = np.array([int(pp_optimizer), int(optim_pose), iters])
    assert len(tl["params"]) == iters and len(tl["grads"]) == iters, (len(tl["params"]), len(tl["grads"]))
    for name in TL_NAMES:   # teacher forcing: the state going into every iteration and the gradients that iteration produced
        key = name if name.startswith("_") else "_" + name
        o[prefix + "_iter_params" + key] = np.stack([p_[name].numpy() for p_ in tl["params"]])
        o[prefix + "_iter_grads" + key] = np.stack([g_[name].numpy() for g_ in tl["grads"]])
    np.savez_compressed(OUT, **o)
    print("added training-loop vectors", prefix, len(o), "arrays; losses", np.round(o[prefix + "_losses"], 5), "views", o[prefix + "_view_uids"])
    return model


out8 = dict(np.load(OUT))
out8["loop_config"] = np.array([TL_V, TL_WM, TL_W, TL_H, TL_ITERS], dtype=np.int64)
out8["loop_cam_w2c"] = np.stack([c.world_view_transform.t().numpy() for c in sc8.cameras])
out8["loop_cam_fov"] = np.array([[c.FoVx, c.FoVy] for c in sc8.cameras])
out8["loop_gt_images"] = np.stack([g_.numpy() for g_ in gts8])
out8["loop_points_noisy"], out8["loop_colors_noisy"], out8["loop_extent"] = pts_noisy, col_noisy, np.float64(sc8.extent)
out8["loop_confidence"], out8["loop_pose_noise_q"], out8["loop_pose_noise_t"] = sc8.confidence.numpy(), sc8.pose_noise_q.numpy(), sc8.pose_noise_t.numpy()
out8["loop_init_scaling_delta"], out8["loop_init_rotation"] = init_scaling_delta.numpy(), init_rotation.numpy()
np.savez_compressed(OUT, **out8)
# second configuration first (plain Adam, poses fixed: reference train.py:98-101,146-147), then the one the scripts run
_run_reference_training("loopb", False, False, 8)
student = _run_reference_training("loop", True, True, TL_ITERS)

# ---- checkpoint contents: GaussianModel.capture() of the reference's own class after the run above (scene/gaussian_model.py:65-80):
# the 13 entries in order, and the optimizer state_dict inside it (per-parameter step / exp_avg / exp_avg_sq, param-group keys)
cap = student.capture()
oc = dict(np.load(OUT))
oc["capture_len"] = np.array(len(cap))
oc["capture_active_sh_degree"] = np.array(cap[0])
for i_, name_ in zip(range(1, 7), ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")):
    oc["capture" + name_] = cap[i_].detach().numpy().copy()
oc["capture_max_radii2D"], oc["capture_xyz_gradient_accum"], oc["capture_denom"] = (cap[7].numpy().copy(), cap[8].numpy().copy(),
                                                                                     cap[9].numpy().copy())
sd_ = cap[10]
oc["capture_opt_group_keys"] = np.array(sorted(sd_["param_groups"][0].keys()))
oc["capture_opt_group_names"] = np.array([g_["name"] for g_ in sd_["param_groups"]])
oc["capture_opt_group_lrs"] = np.array([g_["lr"] for g_ in sd_["param_groups"]], dtype=np.float64)
oc["capture_opt_group_params"] = np.array([g_["params"][0] for g_ in sd_["param_groups"]])
oc["capture_opt_state_ids"] = np.array(sorted(sd_["state"].keys()))
for k_, st_ in sd_["state"].items():
    oc[f"capture_opt_state{k_}_keys"] = np.array(sorted(st_.keys()))
    oc[f"capture_opt_state{k_}_step"] = np.array(int(st_["step"]))
    oc[f"capture_opt_state{k_}_exp_avg"] = st_["exp_avg"].numpy().copy()
    oc[f"capture_opt_state{k_}_exp_avg_sq"] = st_["exp_avg_sq"].numpy().copy()
oc["capture_spatial_lr_scale"] = np.array(float(cap[11]))
oc["capture_P"] = cap[12].detach().numpy().copy()
np.savez_compressed(OUT, **oc)
print("added capture() vectors:", len(oc), "arrays; optimizer groups", list(oc["capture_opt_group_names"]), "state ids", list(oc["capture_opt_state_ids"]))

# ---- test-view pose tracking (reference render.py:99-186, `render_set_optimize`): Gaussians frozen, Adam on (t, q) with
# weight decay and cosine annealing, masked L1, best-loss pose kept — again the reference's own function, taken from its
# file and run around the C oracle operator, on the student the training run above produced.
rsrc = open(os.path.join(REF, "render.py")).read()
rfn = next(n for n in ast.parse(rsrc).body if isinstance(n, ast.FunctionDef) and n.name == "render_set_optimize")
rcode = ast.get_source_segment(rsrc, rfn).replace('device="cuda"', 'device="cpu"').replace(".cuda()", "")
trk = {"poses": [], "losses": [], "saved": []}


class _Tqdm:
    def __init__(self, iterable=None, **k):
        self.it = iterable

    def __iter__(self):
        return iter(self.it)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def update(self, *a, **k):
        pass

    def set_postfix(self, *a, **k):
        pass


def _render_pose_tracked(cam, pc, pipe, bg, camera_pose=None, **k):
    trk["poses"].append(camera_pose.detach().clone())
    return gr.render(cam, pc, pipe, bg, camera_pose=camera_pose, **k)


def _l1_mask_tracked(a, b, m):
    v = loss_utils.l1_loss_mask(a, b, m)
    trk["losses"].append(float(v))
    return v


tv_stub = types.SimpleNamespace(utils=types.SimpleNamespace(save_image=lambda img, path: trk["saved"].append(img.detach().clone())))
TRACK_ITERS = 15
rns = {"os": os, "makedirs": os.makedirs, "tqdm": _Tqdm, "get_tensor_from_camera": pose_utils.get_tensor_from_camera, "torch": torch,
       "render": _render_pose_tracked, "l1_loss_mask": _l1_mask_tracked, "torchvision": tv_stub,
       "args": types.SimpleNamespace(optim_test_pose_iter=TRACK_ITERS, test_fps=False), "perf_counter": __import__("time").perf_counter,
       "json": __import__("json")}
exec(compile(rcode, os.path.join(REF, "render.py"), "exec"), rns)
true_cam = sc8.cameras[1]
w2c_true = true_cam.world_view_transform.t().double()
ang = np.radians(1.5)
dR = torch.tensor([[np.cos(ang), 0.0, np.sin(ang), 0.0], [0.0, 1.0, 0.0, 0.0], [-np.sin(ang), 0.0, np.cos(ang), 0.0], [0.0, 0.0, 0.0, 1.0]])
w2c_guess = dR @ w2c_true
w2c_guess[:3, 3] += torch.tensor([0.03, -0.02, 0.04], dtype=torch.float64)
guess_cam = cm.Camera(colmap_id=9, R=w2c_guess[:3, :3].numpy().T.copy(), T=w2c_guess[:3, 3].numpy().copy(), FoVx=true_cam.FoVx,
                      FoVy=true_cam.FoVy, image=gts8[1], gt_alpha_mask=None, image_name="track", uid=0, data_device="cpu")
with tempfile.TemporaryDirectory() as td:
    rns["render_set_optimize"](td, "test", 12, [guess_cam], student, _Pipe(False, False), bg8)
out9 = dict(np.load(OUT))
out9["track_w2c_guess"], out9["track_gt"], out9["track_iters"] = w2c_guess.float().numpy(), gts8[1].numpy(), np.array(TRACK_ITERS)
out9["track_pose_sequence"] = torch.stack(trk["poses"]).numpy()          # pose used by each of the num_iter renders + the final one
out9["track_losses"] = np.array(trk["losses"], dtype=np.float64)
out9["track_optimal_pose"], out9["track_final_render"] = trk["poses"][-1].numpy(), trk["saved"][0].numpy()
np.savez_compressed(OUT, **out9)
print("added pose-tracking vectors:", len(out9), "arrays; losses", np.round(out9["track_losses"], 5))

# ---- pose export (reference train.py:46-60, `save_pose`): [V,7] poses -> [V,4,4] ordered by COLMAP id
sfn = next(n for n in ast.parse(tsrc).body if isinstance(n, ast.FunctionDef) and n.name == "save_pose")
sns = {"np": np, "torch": torch, "get_camera_from_tensor": pose_utils.get_camera_from_tensor}
exec(compile(ast.Module(body=[sfn], type_ignores=[]), os.path.join(REF, "train.py"), "exec"), sns)
ids10 = [3, 1, 2]
with tempfile.TemporaryDirectory() as td:
    sns["save_pose"](os.path.join(td, "pose_optimized.npy"), student.P, [types.SimpleNamespace(colmap_id=i) for i in ids10])
    saved10 = np.load(os.path.join(td, "pose_optimized.npy"))
out10 = dict(np.load(OUT))
out10["save_pose_in"], out10["save_pose_colmap_ids"], out10["save_pose_out"] = student.P.detach().numpy().copy(), np.array(ids10), saved10
np.savez_compressed(OUT, **out10)
print("added save_pose vectors:", len(out10), "arrays")
