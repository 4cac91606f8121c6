"""`gaussian_renderer.render` with the reference's signature and return dict
(reference gaussian_renderer/__init__.py:23-31,139-144).

InstantSplat's twist (reference :55-59,81-90) is preserved: the rasterizer sees an IDENTITY view
matrix and camera position 0; Gaussian means and (raw, un-normalised) rotations are moved into the
camera frame here from the learnable 7-vector `camera_pose`, so pose gradients flow through autograd.
"""
from __future__ import annotations

import math

import torch

from ..diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from ..fused import pose_activations, sh_features
from ..pose_utils import get_camera_from_tensor, quadmultiply
from ..sh_utils import eval_sh


_IDENTITY = {}


def _identity_view(dev):
    if dev not in _IDENTITY:
        _IDENTITY[dev] = (torch.eye(4, device=dev), torch.zeros(3, device=dev))
    return _IDENTITY[dev]


# False = always take the op-by-op PyTorch glue below (kept for A/B tests against the fused path)
FUSED_GLUE = True


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None, camera_pose=None):
    dev = pc.get_xyz.device
    screenspace_points = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True, device=dev) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass

    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)

    w2c, camera_pos = _identity_view(dev)  # identity view matrix, camera at the origin (reference :55-59)
    projmatrix = viewpoint_camera.projection_matrix  # identity @ projection
    if projmatrix.device != dev:
        projmatrix = projmatrix.to(dev)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=w2c,
        projmatrix=projmatrix, sh_degree=pc.active_sh_degree, campos=camera_pos, prefiltered=False, debug=pipe.debug)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    means2D = screenspace_points
    fused = (not pipe.compute_cov3D_python) and (not pipe.convert_SHs_python) and override_color is None and FUSED_GLUE
    if fused:
        # one HIP launch each way for the pose transform + activations (and the pose-gradient reduction)
        means3D, rot_cam, scales_act, opacity = pose_activations(pc._xyz, pc._rotation, pc._scaling, pc._opacity, camera_pose)
        shs_dc, shs_rest = sh_features(pc)
        rendered_image, radii = rasterizer(means3D=means3D, means2D=means2D, shs=shs_dc, colors_precomp=None,
                                           opacities=opacity, scales=scales_act, rotations=rot_cam, cov3D_precomp=None,
                                           shs_rest=shs_rest)
        return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}

    rel_w2c = get_camera_from_tensor(camera_pose)
    xyz = pc._xyz
    means3D = xyz @ rel_w2c[:3, :3].t() + rel_w2c[:3, 3]
    rot_cam = quadmultiply(camera_pose[:4], pc._rotation)
    opacity = pc.get_opacity

    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales = pc.get_scaling
        rotations = rot_cam

    shs = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = pc.get_xyz - viewpoint_camera.camera_center.to(dev).repeat(pc.get_features.shape[0], 1)
            dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            sh2rgb = eval_sh(pc.active_sh_degree, shs_view, dir_pp_normalized)
            colors_precomp = torch.clamp_min(sh2rgb + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color

    rendered_image, radii = rasterizer(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp,
                                       opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}
