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
from .. import _lib
from .. import diff_gaussian_rasterization as _dgr
from ..fused import pose_activations, render_posed, render_posed_compiled, sh_features
from ..pose_utils import get_camera_from_tensor, quadmultiply
from ..sh_utils import eval_sh


_IDENTITY = {}


def _identity_view(dev):
    if dev not in _IDENTITY:
        _IDENTITY[dev] = (torch.eye(4, device=dev), torch.zeros(3, device=dev))
    return _IDENTITY[dev]


_ZEROS = {}


def _zero_leaf(like):
    key = (tuple(like.shape), like.device, like.dtype)
    base = _ZEROS.get(key)
    if base is None:
        if len(_ZEROS) > 16:
            _ZEROS.clear()
        base = _ZEROS[key] = torch.zeros(like.shape, dtype=like.dtype, device=like.device)
    return base.detach().requires_grad_(True)


# "posed" (default): one autograd node for the whole render body (fused.render_posed); True: the round-1 fused glue (pose /
# activation kernel + SH view + operator, three nodes); False: the op-by-op PyTorch glue below.  The latter two are kept
# for A/B tests: all three must give the same image and gradients.
FUSED_GLUE = "posed"


def _operator_inputs(viewpoint_camera, pc, pipe, camera_pose, scaling_modifier, override_color, dev):
    """Op-by-op glue (PyTorch autograd): keyword arguments for the rasterizer, Gaussians moved into the camera frame.
    Semantics of reference gaussian_renderer/__init__.py:81-122, including the two python-flag variants."""
    w2c_rel = get_camera_from_tensor(camera_pose)
    kw = {"means3D": pc._xyz @ w2c_rel[:3, :3].t() + w2c_rel[:3, 3], "opacities": pc.get_opacity}
    if pipe.compute_cov3D_python:
        kw["cov3D_precomp"] = pc.get_covariance(scaling_modifier)
    else:
        kw["scales"] = pc.get_scaling
        kw["rotations"] = quadmultiply(camera_pose[:4], pc._rotation)   # raw Hamilton product, nothing normalised
    if override_color is not None:
        kw["colors_precomp"] = override_color
    elif pipe.convert_SHs_python:
        n_coeff = (pc.max_sh_degree + 1) ** 2
        coeffs = pc.get_features.transpose(1, 2).view(-1, 3, n_coeff)
        rays = pc.get_xyz - viewpoint_camera.camera_center.to(dev)      # world-frame centre against world-frame means, as upstream
        kw["colors_precomp"] = torch.clamp_min(eval_sh(pc.active_sh_degree, coeffs, rays / rays.norm(dim=1, keepdim=True)) + 0.5, 0.0)
    else:
        kw["shs"] = pc.get_features
    return kw


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None, camera_pose=None):
    xyz = pc.get_xyz
    dev = xyz.device
    # zero tensor whose .grad receives the screen-space mean gradients (the reference's `viewspace_points`, :39-48).  The
    # reference makes it a non-leaf (`zeros + 0`) and asks autograd to retain its gradient; a leaf gets the same .grad
    # without the extra elementwise kernel and the two autograd nodes.  Its VALUE is never read by the operator, so every
    # call hands out a fresh leaf over one shared all-zero buffer per (shape, device) instead of filling a new one.
    screenspace_points = _zero_leaf(xyz)

    view_identity, origin = _identity_view(dev)      # reference :55-59: identity view matrix, camera at the origin
    projmatrix = viewpoint_camera.projection_matrix  # identity @ projection
    if projmatrix.device != dev:
        projmatrix = projmatrix.to(dev)
    default_pipeline = override_color is None and not (pipe.compute_cov3D_python or pipe.convert_SHs_python)
    if FUSED_GLUE == "posed" and default_pipeline and pc.max_sh_degree == 3 and not pipe.debug and not _dgr._KEEP_LAST_FRAME:
        ext = _lib.compiled()
        if ext is not None:
            # the compiled node takes the settings as plain arguments: no settings tuple is built on this path (the training
            # loop's path: everything between the optimizer step and the first launch of the next frame is GPU idle time)
            fx, fy = viewpoint_camera.FoVx, viewpoint_camera.FoVy
            tans = viewpoint_camera.__dict__.get("_gs_tanfov")
            if tans is None or tans[0] != fx or tans[1] != fy:
                tans = viewpoint_camera.__dict__["_gs_tanfov"] = (fx, fy, math.tan(fx * 0.5), math.tan(fy * 0.5))
            image, radii, visible = render_posed_compiled(ext, pc, camera_pose, screenspace_points, bg_color, view_identity, projmatrix,
                                                          origin, int(viewpoint_camera.image_height), int(viewpoint_camera.image_width),
                                                          tans[2], tans[3], float(scaling_modifier), int(pc.active_sh_degree))
            # (`visible` is the reference's `radii > 0`, written by the projection kernel: no elementwise launch behind the node)
            return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": visible, "radii": radii}
    settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=view_identity, projmatrix=projmatrix, sh_degree=pc.active_sh_degree,
        campos=origin, prefiltered=False, debug=pipe.debug)

    if FUSED_GLUE == "posed" and default_pipeline and pc.max_sh_degree == 3:
        # the whole differentiable body as one autograd node: the projection kernels take the raw parameters + the pose
        # (no GaussianRasterizer module is instantiated on this path: constructing an nn.Module costs ~10 us per call)
        image, radii = render_posed(pc, camera_pose, screenspace_points, settings)
    elif FUSED_GLUE and default_pipeline:
        # one HIP launch each way for the pose transform + activations (and the pose-gradient reduction)
        means3D, rot_cam, scales_act, opacity = pose_activations(pc._xyz, pc._rotation, pc._scaling, pc._opacity, camera_pose)
        shs_dc, shs_rest = sh_features(pc)
        image, radii = GaussianRasterizer(raster_settings=settings)(
            means3D=means3D, means2D=screenspace_points, shs=shs_dc, opacities=opacity, scales=scales_act, rotations=rot_cam,
            shs_rest=shs_rest)
    else:
        image, radii = GaussianRasterizer(raster_settings=settings)(
            means2D=screenspace_points,
            **_operator_inputs(viewpoint_camera, pc, pipe, camera_pose, scaling_modifier, override_color, dev))
    return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}
