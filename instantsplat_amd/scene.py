"""Gaussian state for the train/render hot path: tensor names, layouts, activations, initialisation
and optimiser groups of the reference's GaussianModel (reference scene/gaussian_model.py:29-243),
minus everything that is never executed by InstantSplat (densify/prune, :328-477)."""
from __future__ import annotations


import torch
import torch.nn as nn

from . import _lib
from .optim import PerPointAdam, get_expon_lr_func
from .pose_utils import get_tensor_from_camera
from .sh_utils import RGB2SH
from .simple_knn._C import distCUDA2


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def build_covariance_from_scaling_rotation(scaling, scaling_modifier, rotation):
    """Sigma = R S S^T R^T packed [xx,xy,xz,yy,yz,zz]; R from the NORMALISED quaternion
    (reference scene/gaussian_model.py:32-36 + utils/general_utils.py:64-110)."""
    q = rotation / rotation.norm(dim=1, keepdim=True)
    r, x, y, z = q.unbind(dim=1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    L = R * (scaling_modifier * scaling)[:, None, :]
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)


class GaussianModel:
    def __init__(self, sh_degree: int):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        e = torch.empty(0)
        self._xyz = self._features_dc = self._features_rest = self._scaling = self._rotation = self._opacity = e
        self.optimizer = None
        self.spatial_lr_scale = 0
        self.P = None
        self.test_P = None   # (reference: only ever read, by get_RT_test)
        # densification statistics of the reference's model (scene/gaussian_model.py:55-57).  Densification is disabled in
        # InstantSplat (train.py:195-206), so they only exist to keep capture() / restore() tuples interchangeable.
        self.max_radii2D = self.xyz_gradient_accum = self.denom = e

    # ---- checkpoint tuple (reference :65-99): same 13 entries in the same order, so a chkpnt<iteration>.pth written by
    # either side loads on the other
    def capture(self):
        return (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation,
                self._opacity, self.max_radii2D, self.xyz_gradient_accum, self.denom, self.optimizer.state_dict(),
                self.spatial_lr_scale, self.P)

    def restore(self, model_args, training_args, confidence_lr=None):
        """Reference :82-99.  Like the reference it rebuilds the optimizer with `training_setup` — which on a run started
        with --pp_optimizer silently drops the per-point multiplier (the state dict restores moments and steps, not the
        optimizer class); pass `confidence_lr` to rebuild the per-point optimizer instead (not in the reference)."""
        (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity,
         self.max_radii2D, xyz_gradient_accum, denom, opt_dict, self.spatial_lr_scale, self.P) = model_args
        if confidence_lr is not None:
            self.training_setup_pp(training_args, confidence_lr)
        else:
            self.training_setup(training_args)
        self.xyz_gradient_accum, self.denom = xyz_gradient_accum, denom
        opt_dict = dict(opt_dict, param_groups=[dict(g) for g in opt_dict["param_groups"]])
        if confidence_lr is None:
            for g in opt_dict["param_groups"]:       # plain Adam has no multiplier: do not let the saved groups re-introduce it
                g["per_point_lr"] = None
        self.optimizer.load_state_dict(opt_dict)

    # ---- PLY (reference :247-326; io_formats writes / reads the same binary layout without `plyfile`)
    def save_ply(self, path):
        import os
        from .io_formats import save_gaussian_ply
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        save_gaussian_ply(path, self)

    def load_ply(self, path, device="cuda"):
        from .io_formats import load_gaussian_ply
        for k, v in load_gaussian_ply(path, self.max_sh_degree, device).items():
            setattr(self, k, nn.Parameter(v.requires_grad_(True)))
        self.active_sh_degree = self.max_sh_degree

    # ---- activations (reference :101-124)
    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    def get_covariance(self, scaling_modifier=1):
        return build_covariance_from_scaling_rotation(self.get_scaling, scaling_modifier, self._rotation)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ---- poses (reference :126-136)
    def init_RT_seq(self, cameras, device):
        poses = [get_tensor_from_camera(cam.world_view_transform.transpose(0, 1).cpu()) for cam in cameras]
        self.P = torch.stack(poses).to(device).requires_grad_(True)

    POSE_ROW_NODE = True   # A/B switch: False = plain indexing (autograd's select-backward builds the table gradient)

    def get_RT(self, idx):
        """Row `idx` of the learnable pose table (reference :134-136).  Through the compiled binding it is a node of its own whose
        backward receives the whole table's gradient from the render node's last kernel (csrc_torch/binding.cpp PoseRowFn)
        instead of building it with a fill and a copy; values and gradients are those of `self.P[idx]`."""
        if self.POSE_ROW_NODE and type(idx) is int and self.P.is_contiguous():
            ext = _lib.compiled()
            if ext is not None:
                return ext.pose_row(self.P, idx)
        return self.P[idx]

    def get_RT_test(self, idx):
        """reference :138-140 (read by train.py:276 for the test cameras of `training_report`): row `idx` of `test_P`, a table the
        reference never fills itself — whoever evaluates test views assigns it first (render.py optimises its own copy)."""
        return self.test_P[idx]

    # ---- initialisation from a point cloud (reference :146-172)
    def create_from_pcd(self, points: torch.Tensor, colors: torch.Tensor, spatial_lr_scale: float, device, scale_gaussian=None):
        """scale_gaussian: optional per-point cap on the initial scale (reference :157-159, `--init_scale_from_view_depth`)."""
        self.spatial_lr_scale = spatial_lr_scale
        pts = torch.as_tensor(points).float().to(device).contiguous()   # (a [3,N] array's transpose arrives with its strides)
        fused_color = RGB2SH(torch.as_tensor(colors).float().to(device))
        n = pts.shape[0]
        features = torch.zeros((n, 3, (self.max_sh_degree + 1) ** 2), dtype=torch.float32, device=device)
        features[:, :3, 0] = fused_color
        dist2 = torch.clamp_min(distCUDA2(pts), 0.0000001)
        if scale_gaussian is not None:
            dist2 = torch.min(torch.as_tensor(scale_gaussian ** 2).float().to(device), dist2)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
        rots = torch.zeros((n, 4), device=device)
        rots[:, 0] = 1
        opacities = inverse_sigmoid(0.1 * torch.ones((n, 1), dtype=torch.float, device=device))
        self._xyz = nn.Parameter(pts.requires_grad_(True))
        self._features_dc = nn.Parameter(features[:, :, 0:1].transpose(1, 2).contiguous().requires_grad_(True))
        self._features_rest = nn.Parameter(features[:, :, 1:].transpose(1, 2).contiguous().requires_grad_(True))
        self._scaling = nn.Parameter(scales.requires_grad_(True))
        self._rotation = nn.Parameter(rots.requires_grad_(True))
        self._opacity = nn.Parameter(opacities.requires_grad_(True))
        self.max_radii2D = torch.zeros((n,), device=device)

    # ---- optimiser (reference :173-243)
    def _groups(self, o, per_point_lr):
        xyz = {"params": [self._xyz], "lr": o.position_lr_init * self.spatial_lr_scale, "name": "xyz"}
        if per_point_lr is not None:
            xyz["per_point_lr"] = per_point_lr
        return [xyz,
                {"params": [self._features_dc], "lr": o.feature_lr * 10, "name": "f_dc"},
                {"params": [self._features_rest], "lr": o.feature_lr / 20.0 * 10, "name": "f_rest"},
                {"params": [self._opacity], "lr": o.opacity_lr, "name": "opacity"},
                {"params": [self._scaling], "lr": o.scaling_lr * 10, "name": "scaling"},
                {"params": [self._rotation], "lr": o.rotation_lr * 10, "name": "rotation"},
                {"params": [self.P], "lr": o.rotation_lr * 0.1, "name": "pose"}]

    def _schedulers(self, o):
        n = self._xyz.shape[0]   # reference :175-176 / :205-206 (statistics of the disabled densification, kept for capture())
        self.xyz_gradient_accum = torch.zeros((n, 1), device=self._xyz.device)
        self.denom = torch.zeros((n, 1), device=self._xyz.device)
        self.xyz_scheduler_args = get_expon_lr_func(o.position_lr_init * self.spatial_lr_scale,
                                                    o.position_lr_final * self.spatial_lr_scale,
                                                    lr_delay_mult=o.position_lr_delay_mult, max_steps=o.position_lr_max_steps)
        self.cam_scheduler_args = get_expon_lr_func(o.rotation_lr * 0.1, o.rotation_lr * 0.001,
                                                    lr_delay_mult=o.position_lr_delay_mult, max_steps=o.iterations)

    def training_setup(self, o):
        """Plain Adam (eps 1e-15) over the same 7 groups — the reference's non --pp_optimizer path (:173-201);
        expressed with PerPointAdam without a multiplier so the step still runs in the fused HIP kernel."""
        self.optimizer = PerPointAdam(self._groups(o, None), lr=0.0, betas=(0.9, 0.999), eps=1e-15)
        self._schedulers(o)

    def training_setup_pp(self, o, confidence_lr=None):
        self.per_point_lr = confidence_lr
        self.optimizer = PerPointAdam(self._groups(o, confidence_lr), lr=0, betas=(0.9, 0.999), eps=1e-15, weight_decay=0.0)
        self._schedulers(o)

    def update_learning_rate(self, iteration):
        for g in self.optimizer.param_groups:
            if g["name"] == "pose":
                g["lr"] = self.cam_scheduler_args(iteration)
            if g["name"] == "xyz":
                g["lr"] = self.xyz_scheduler_args(iteration)


def confidence_to_lr_modifiers(confidence: torch.Tensor, scale=(1.0, 100.0)) -> torch.Tensor:
    """MASt3R confidence -> per-point LR multiplier (reference train.py:63-85, called with scale=(1,100) at :96)."""
    inv = 1.0 - torch.sigmoid(confidence.float())
    lo, hi = scale
    return inv * (hi - lo) + lo
