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
from . import diff_gaussian_rasterization as dgr


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
        with _lib.on_device(dev):
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
        with _lib.on_device(dev):
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


class _RenderPosed(torch.autograd.Function):
    """render()'s whole differentiable body as ONE autograd node (mi355gs_posed_forward_preprocess / _posed_backward): raw
    GaussianModel tensors + the 7-vector camera pose in, image out; the projection kernels apply the camera-frame transform
    and the activations themselves.  Replaces the three nodes (pose/activations, SH view, rasterizer) of the round-1 fused
    glue: three launches and ~0.1 ms of Python / autograd-engine time fewer per iteration on the drop-in path, and none
    of the camera-frame intermediates (means, rotations, scales, opacities: 44 B/Gaussian each way) exist in HBM."""

    @staticmethod
    def forward(ctx, xyz, rot, scaling, opacity_logit, f_dc, f_rest, pose, means2D, settings):
        s = settings
        L = _lib.lib()
        (xyz, rot, scaling, opl, f_dc, f_rest, pose, bg, view, proj, origin), dev = _lib.f32c_on_one_device(
            xyz, rot, scaling, opacity_logit, f_dc, f_rest, pose, s.bg, s.viewmatrix, s.projmatrix, s.campos)
        P, D = xyz.shape[0], int(s.sh_degree)
        H, W = int(s.image_height), int(s.image_width)
        stream, debug = _lib.stream_ptr(dev), (1 if s.debug else 0)
        radii, color, geom, tiles, num_rendered = dgr.frame_buffers(L, P, W, H, dev)
        with _lib.on_device(dev):
            _lib.check(L.mi355gs_posed_forward_preprocess(
                stream, P, D, W, H, _lib.ptr(xyz), _lib.ptr(f_dc), _lib.ptr(f_rest), _lib.ptr(opl), _lib.ptr(scaling),
                float(s.scale_modifier), _lib.ptr(rot), _lib.ptr(pose), _lib.ptr(view), _lib.ptr(proj), _lib.ptr(origin),
                float(s.tanfovx), float(s.tanfovy), _lib.ptr(radii), _lib.ptr(geom), _lib.ptr(tiles), _lib.ptr(num_rendered), None, None,
                debug), "posed_forward_preprocess")
            R, binning = dgr.size_and_render(L, stream, dev, P, W, H, num_rendered, bg, geom, tiles, color, debug)
        if dgr._KEEP_LAST_FRAME:
            dgr._LAST_FRAME.update(tiles=tiles, W=W, H=H, geom=geom, binning=binning, capacity=int(R), P=P)
        ctx.settings, ctx.capacity, ctx.dims = s, R, (P, D, W, H)
        ctx.save_for_backward(xyz, rot, scaling, opl, f_dc, f_rest, pose, radii, geom, tiles, binning, bg, view, proj, origin, color)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_color, _grad_radii):
        s, L = ctx.settings, _lib.lib()
        P, D, W, H = ctx.dims
        xyz, rot, scaling, opl, f_dc, f_rest, pose, radii, geom, tiles, binning, bg, view, proj, origin, color = ctx.saved_tensors
        dev = xyz.device
        g = _lib.f32c(grad_color)
        _lib.require_device(g)
        new = lambda like: torch.empty_like(like)
        d_xyz, d_rot, d_scaling, d_opl, d_fdc, d_m2d = new(xyz), new(rot), new(scaling), new(opl), new(f_dc), new(xyz)
        # below its SH degree f_rest gets the all-zero gradient cat(f_dc, f_rest) would give it (the optimizer then takes
        # PerPointAdam's zero-gradient step on it, as in the reference)
        d_frest = torch.zeros_like(f_rest) if D == 0 else new(f_rest)
        d_pose = torch.empty(7, dtype=torch.float32, device=dev)
        scratch = dgr._empty_bytes(dgr.grad_scratch_bytes(L, P), dev)
        pose_scratch = torch.empty(16 * ((P + 255) // 256) + 32, dtype=torch.float32, device=dev)
        dgr.check_frame_buffers(L, binning, int(ctx.capacity), W, H)
        with _lib.on_device(dev):
            _lib.check(L.mi355gs_posed_backward(
                _lib.stream_ptr(dev), P, D, W, H, _lib.ptr(bg), _lib.ptr(xyz), _lib.ptr(f_dc), _lib.ptr(f_rest), _lib.ptr(opl),
                _lib.ptr(scaling), float(s.scale_modifier), _lib.ptr(rot), _lib.ptr(pose), _lib.ptr(view), _lib.ptr(proj),
                _lib.ptr(origin), float(s.tanfovx), float(s.tanfovy), _lib.ptr(geom), _lib.ptr(tiles), _lib.ptr(binning),
                int(ctx.capacity), _lib.ptr(radii), _lib.ptr(color), _lib.ptr(g), _lib.ptr(scratch), _lib.ptr(pose_scratch),
                _lib.ptr(d_xyz), _lib.ptr(d_m2d), _lib.ptr(d_fdc), _lib.ptr(d_frest) if D else None, _lib.ptr(d_opl),
                _lib.ptr(d_scaling), _lib.ptr(d_rot), _lib.ptr(d_pose), 0, 0, 0, 1 if s.debug else 0), "posed_backward")
        return d_xyz, d_rot, d_scaling, d_opl, d_fdc, d_frest, d_pose, d_m2d, None


_LAST_COUNT = {}   # (P, W, H, hint key) -> instance count of the last frame like this one (sizes the speculative stage 2)


def render_posed_compiled(ext, pc, pose, means2D, bg, view, proj, origin, H, W, tanfovx, tanfovy, scale_modifier, degree):
    """The compiled node (csrc_torch/binding.cpp) with the BinningPolicy bookkeeping of `size_and_render` around it.
    -> [image, radii, visible]: `visible` = radii > 0 as a bool tensor the projection kernel wrote (no compare kernel)."""
    policy = dgr.BinningPolicy
    xyz = pc._xyz
    dev = xyz.device
    slot = dgr.count_slot(dev)
    cap = policy.deferred_capacity()
    key = policy.current_key
    if cap is not None:
        out = ext.render_posed(xyz, pc._rotation, pc._scaling, pc._opacity, pc._features_dc, pc._features_rest, pose, means2D,
                               bg, view, proj, origin, H, W, tanfovx, tanfovy, scale_modifier, degree, cap, 0, slot)
        policy.defer(slot, cap, dev)
        return out
    # exact mode: the reference operator's blocking count read-back; the count of the previous frame like this one lets the
    # node enqueue stage 2 before the count of THIS frame has arrived (see RenderPosedFn::forward)
    ck = (xyz.shape[0], W, H, key)
    out = ext.render_posed(xyz, pc._rotation, pc._scaling, pc._opacity, pc._features_dc, pc._features_rest, pose, means2D,
                           bg, view, proj, origin, H, W, tanfovx, tanfovy, scale_modifier, degree, -1, _LAST_COUNT.get(ck, 0), slot)
    r = dgr.read_count(slot)
    if len(_LAST_COUNT) > 256:
        _LAST_COUNT.clear()
    _LAST_COUNT[ck] = r
    if key is not None:
        policy.known[key] = r
    return out


def render_posed(pc, pose, means2D, settings):
    """-> (image[3,H,W], radii[P]) for the default pipeline (SH colours of the active degree, scale/rotation covariance).

    The node is the compiled one (csrc_torch/binding.cpp: same two C-ABI calls each way, issued from C++) unless the
    binding is switched to ctypes, the operator's debug mode is on (snapshot dumps are written by the Python node) or
    bench.py's frame-statistics hook wants the frame's scratch kept."""
    s = settings
    ext = None if (s.debug or dgr._KEEP_LAST_FRAME) else _lib.compiled()
    if ext is None:
        with dgr.backward_follows(pc._xyz, pc._rotation, pc._scaling, pc._opacity, pc._features_dc, pc._features_rest, pose, means2D):
            return _RenderPosed.apply(pc._xyz, pc._rotation, pc._scaling, pc._opacity, pc._features_dc, pc._features_rest, pose, means2D,
                                      settings)
    return render_posed_compiled(ext, pc, pose, means2D, s.bg, s.viewmatrix, s.projmatrix, s.campos, int(s.image_height),
                                 int(s.image_width), float(s.tanfovx), float(s.tanfovy), float(s.scale_modifier), int(s.sh_degree))[:2]


def sh_features(pc):
    """-> (shs, shs_rest) for GaussianRasterizer.forward without materialising cat(f_dc, f_rest)."""
    if pc.active_sh_degree == 0:
        return _SHDegree0View.apply(pc._features_dc, pc._features_rest), None
    return pc._features_dc, pc._features_rest
