"""Fused pieces of the InstantSplat render glue.

`pose_activations` is one HIP launch each way for what the reference does with ~60 eager PyTorch
kernels per render (reference gaussian_renderer/__init__.py:81-103): world->camera transform of the
means from the learnable 7-vector pose, Hamilton product of the (raw) pose quaternion with the (raw)
Gaussian quaternions, sigmoid / exp activations — and, in backward, the autograd of all of it
including the reduction over every Gaussian to the seven pose gradients.
"""
from __future__ import annotations

import torch

from . import _lib


class _PoseActivations(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, rot, scaling, opacity_logit, pose):
        L = _lib.lib()
        xyz, rot, scaling, opl, pose = map(_lib.f32c, (xyz, rot, scaling, opacity_logit, pose))
        dev = _lib.require_device(xyz, rot, scaling, opl, pose)
        P = xyz.shape[0]
        means = torch.empty_like(xyz)
        rot_cam = torch.empty_like(rot)
        scales = torch.empty_like(scaling)
        opac = torch.empty_like(opl)
        _lib.check(L.mi355gs_pose_forward(_lib.stream_ptr(dev), P, _lib.ptr(xyz), _lib.ptr(rot), _lib.ptr(scaling), _lib.ptr(opl),
                                          _lib.ptr(pose), _lib.ptr(means), _lib.ptr(rot_cam), _lib.ptr(scales), _lib.ptr(opac)),
                   "pose_forward")
        ctx.save_for_backward(xyz, rot, scales, opac, pose)
        return means, rot_cam, scales, opac

    @staticmethod
    def backward(ctx, g_means, g_rot, g_scales, g_opac):
        L = _lib.lib()
        xyz, rot, scales, opac, pose = ctx.saved_tensors
        dev = xyz.device
        P = xyz.shape[0]
        z = lambda g, like: torch.zeros_like(like) if g is None else _lib.f32c(g)
        g_means, g_rot, g_scales, g_opac = z(g_means, xyz), z(g_rot, rot), z(g_scales, scales), z(g_opac, opac)
        d_xyz, d_rot, d_scaling, d_opl = torch.empty_like(xyz), torch.empty_like(rot), torch.empty_like(scales), torch.empty_like(opac)
        d_pose = torch.empty(7, dtype=torch.float32, device=dev)
        scratch = torch.empty(32, dtype=torch.float32, device=dev)
        _lib.check(L.mi355gs_pose_backward(_lib.stream_ptr(dev), P, _lib.ptr(xyz), _lib.ptr(rot), _lib.ptr(scales), _lib.ptr(opac),
                                           _lib.ptr(pose), _lib.ptr(g_means), _lib.ptr(g_rot), _lib.ptr(g_scales), _lib.ptr(g_opac),
                                           _lib.ptr(d_xyz), _lib.ptr(d_rot), _lib.ptr(d_scaling), _lib.ptr(d_opl), _lib.ptr(d_pose),
                                           _lib.ptr(scratch)), "pose_backward")
        return d_xyz, d_rot, d_scaling, d_opl, d_pose


def pose_activations(xyz, rot, scaling, opacity_logit, pose):
    """-> (means_cam[P,3], rot_cam[P,4], scales[P,3], opacity[P,1])"""
    return _PoseActivations.apply(xyz, rot, scaling, opacity_logit, pose)


class _SHDegree0View(torch.autograd.Function):
    """`get_features` at SH degree 0 without materialising cat(f_dc, f_rest) (192 B/Gaussian written and read per
    render, reference scene/gaussian_model.py:114-117): the rasterizer reads only coefficient 0, so f_dc is handed
    over as an [P,1,3] SH tensor.  Backward still gives f_rest the all-zero gradient the cat would give it, so the
    optimizer sees exactly the reference's gradients (PerPointAdam then runs its zero-gradient step on f_rest)."""

    @staticmethod
    def forward(ctx, f_dc, f_rest):
        ctx.rest_shape = f_rest.shape
        ctx.rest_device = f_rest.device
        return f_dc.view_as(f_dc)

    @staticmethod
    def backward(ctx, g):
        return g, torch.zeros(ctx.rest_shape, dtype=torch.float32, device=ctx.rest_device)


def sh_features(pc):
    """-> (shs, shs_rest) for GaussianRasterizer.forward without materialising cat(f_dc, f_rest)."""
    if pc.active_sh_degree == 0:
        return _SHDegree0View.apply(pc._features_dc, pc._features_rest), None
    return pc._features_dc, pc._features_rest
