"""ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED (see below).

Dense PyTorch restatement (CPU, fp32 or fp64, autograd-differentiable) of the differentiable
3D-Gaussian rasterizer the reference calls through
``GaussianRasterizer(raster_settings)(means3D=, means2D=, shs=, colors_precomp=, opacities=,
scales=, rotations=, cov3D_precomp=)`` at reference ``gaussian_renderer/__init__.py:126-135``.

The arithmetic of that operator lives in graphdeco-inria/diff-gaussian-rasterization, an
UN-VENDORED submodule with no pinned revision (reference ``.gitmodules:4-6``; directory empty), and
the reference ships no tests or golden vectors, so this oracle restates the published algorithm
 and is anchored on the in-tree pieces that do exist:
  * SH basis / constants      -> reference ``utils/sh_utils.py:24-117`` (golden: tests/golden)
  * cov3D = R S S^T R^T, packed [xx,xy,xz,yy,yz,zz], quaternion (w,x,y,z)
                              -> reference ``utils/general_utils.py:64-110``, ``scene/gaussian_model.py:32-36``
  * projection conventions    -> reference ``utils/graphics_utils.py:71-91``, ``scene/cameras.py:48-57``
"PARITY UNPINNED": no output of the reference CUDA rasterizer is available to check against.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

It is O(pixels x Gaussians) in memory: use it for small cases (<= ~128x128 px, <= a few thousand
Gaussians).  oracle/gs_ref.c is the tile-based C restatement for full-size inputs; the two are
checked against each other in tests/test_oracle.py.
"""
from __future__ import annotations

from typing import NamedTuple

import torch

TILE = 16

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


class RasterSettings(NamedTuple):
    """Same 12 fields, same order, as the settings tuple built at reference
    gaussian_renderer/__init__.py:60-76."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def sh_to_rgb(deg: int, shs: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """shs [P,M,3] (coefficient-major, as the kernel receives them), dirs [P,3] unit.
    Basis order/signs follow reference utils/sh_utils.py:74-100. Returns [P,3] before +0.5/clamp."""
    res = SH_C0 * shs[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * shs[:, 6] + SH_C2[3] * xz * shs[:, 7]
                   + SH_C2[4] * (xx - yy) * shs[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3.0 * xx - yy) * shs[:, 9] + SH_C3[1] * xy * z * shs[:, 10]
                       + SH_C3[2] * y * (4.0 * zz - xx - yy) * shs[:, 11]
                       + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * shs[:, 12]
                       + SH_C3[4] * x * (4.0 * zz - xx - yy) * shs[:, 13]
                       + SH_C3[5] * z * (xx - yy) * shs[:, 14] + SH_C3[6] * x * (xx - 3.0 * yy) * shs[:, 15])
    return res


def quat_to_rot_raw(q: torch.Tensor) -> torch.Tensor:
    """R(q) for q=(r,x,y,z) used AS GIVEN (no normalisation) — same polynomial as reference
    utils/general_utils.py:90-98 minus its normalisation step (the kernel does not normalise,
    SURVEY.md §0.5)."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(-1, 3, 3)


SCALE_GRAD_EXACT = False   # True: autograd's own d/dscale of Sigma(mod * scale); False: the published operator's dL/d(mod * scale)


def cov3d_from_scale_rot(scales, mod, rots):
    """Sigma = R S S^T R^T packed [xx,xy,xz,yy,yz,zz] (layout of reference
    utils/general_utils.py:64-76).  The gradient that reaches `scales` follows the published operator unless SCALE_GRAD_EXACT:
    value mod * scale, derivative 1 (the operator hands back dL/d(mod * scale) as dL/dscale; see oracle/gs_ref.c)."""
    R = quat_to_rot_raw(rots)
    s = mod * scales if (SCALE_GRAD_EXACT or mod == 1.0) else (mod * scales).detach() + (scales - scales.detach())
    L = R * s[:, None, :]
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=-1)


class _ConicInverse(torch.autograd.Function):
    """conic = (c, -b, a)/det with the upstream backward's 1/(det^2 + 1e-7)."""

    @staticmethod
    def forward(ctx, a, b, c):
        det = a * c - b * b
        ctx.save_for_backward(a, b, c, det)
        inv = 1.0 / det
        return c * inv, -b * inv, a * inv

    @staticmethod
    def backward(ctx, gA, gB, gC):
        a, b, c, det = ctx.saved_tensors
        D = 1.0 / (det * det + 1e-7)
        ga = D * (-c * c * gA + b * c * gB + (det - a * c) * gC)
        gc = D * ((det - a * c) * gA + a * b * gB - a * a * gC)
        gb = D * (2 * b * c * gA - (det + 2 * b * b) * gB + 2 * a * b * gC)
        return ga, gb, gc


def _tp43(p, m):
    """transformPoint4x3 with a 16-vector m in the reference's transposed (column-major flat) storage."""
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    return torch.stack([m[0] * x + m[4] * y + m[8] * z + m[12],
                        m[1] * x + m[5] * y + m[9] * z + m[13],
                        m[2] * x + m[6] * y + m[10] * z + m[14]], dim=-1)


def _tp44(p, m):
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    return torch.stack([m[0] * x + m[4] * y + m[8] * z + m[12],
                        m[1] * x + m[5] * y + m[9] * z + m[13],
                        m[2] * x + m[6] * y + m[10] * z + m[14],
                        m[3] * x + m[7] * y + m[11] * z + m[15]], dim=-1)


def preprocess(means3D, opacities, settings: RasterSettings, shs=None, colors_precomp=None, scales=None,
               rotations=None, cov3D_precomp=None, means2D=None):
    """SURVEY.md A.1.  Returns dict of per-Gaussian tensors (all P long; `visible` masks the culled)."""
    dt = means3D.dtype
    P = means3D.shape[0]
    W, H = settings.image_width, settings.image_height
    view = settings.viewmatrix.to(dt).reshape(-1)
    proj = settings.projmatrix.to(dt).reshape(-1)
    campos = settings.campos.to(dt).reshape(-1)
    fx = W / (2.0 * settings.tanfovx)
    fy = H / (2.0 * settings.tanfovy)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE

    p_view = _tp43(means3D, view)
    in_front = p_view[:, 2] > 0.2
    p_hom = _tp44(means3D, proj)
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    p_proj = p_hom[:, :3] * p_w[:, None]

    if cov3D_precomp is not None:
        cov3D = cov3D_precomp
    else:
        cov3D = cov3d_from_scale_rot(scales, settings.scale_modifier, rotations)

    # EWA: clamp is a constant w.r.t. the clamped coordinate (A.4 #2)
    tz = p_view[:, 2]
    tz_safe = torch.where(in_front, tz, torch.ones_like(tz))
    limx, limy = 1.3 * settings.tanfovx, 1.3 * settings.tanfovy
    txtz, tytz = p_view[:, 0] / tz_safe, p_view[:, 1] / tz_safe
    in_x = (txtz >= -limx) & (txtz <= limx)
    in_y = (tytz >= -limy) & (tytz <= limy)
    tx = torch.where(in_x, p_view[:, 0], (txtz.clamp(-limx, limx) * tz_safe).detach())
    ty = torch.where(in_y, p_view[:, 1], (tytz.clamp(-limy, limy) * tz_safe).detach())
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz_safe, zero, -(fx * tx) / (tz_safe * tz_safe),
                     zero, fy / tz_safe, -(fy * ty) / (tz_safe * tz_safe)], dim=-1).reshape(P, 2, 3)
    Wm = torch.stack([view[0], view[4], view[8], view[1], view[5], view[9], view[2], view[6], view[10]]).reshape(3, 3)
    Sig = torch.stack([cov3D[:, 0], cov3D[:, 1], cov3D[:, 2], cov3D[:, 1], cov3D[:, 3], cov3D[:, 4],
                       cov3D[:, 2], cov3D[:, 4], cov3D[:, 5]], dim=-1).reshape(P, 3, 3)
    M = J @ Wm
    cov2 = M @ Sig @ M.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    det_ok = det != 0
    a_s = torch.where(det_ok, a, torch.ones_like(a))
    b_s = torch.where(det_ok, b, torch.zeros_like(b))
    c_s = torch.where(det_ok, c, torch.ones_like(c))
    cA, cB, cC = _ConicInverse.apply(a_s, b_s, c_s)
    mid = 0.5 * (a + c)
    lam1 = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    lam2 = mid - torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(lam1, lam2))).detach()
    px = ((p_proj[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((p_proj[:, 1] + 1.0) * H - 1.0) * 0.5
    if means2D is not None:  # gradient sink in the reference's NDC-scaled units (A.4 #6)
        px = px + means2D[:, 0] * (0.5 * W)
        py = py + means2D[:, 1] * (0.5 * H)

    def _trunc_clamp(v, hi):
        return torch.clamp(torch.trunc(v).to(torch.int64), 0, hi)

    pxd, pyd = px.detach(), py.detach()
    rminx = _trunc_clamp((pxd - radius) / TILE, gx)
    rminy = _trunc_clamp((pyd - radius) / TILE, gy)
    rmaxx = _trunc_clamp((pxd + radius + TILE - 1) / TILE, gx)
    rmaxy = _trunc_clamp((pyd + radius + TILE - 1) / TILE, gy)
    tiles_touched = (rmaxx - rminx) * (rmaxy - rminy)
    visible = in_front & det_ok & (tiles_touched > 0)

    if colors_precomp is not None:
        rgb = colors_precomp
        clamped = torch.zeros(P, 3, dtype=torch.bool)
    else:
        d = means3D - campos[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        raw = sh_to_rgb(settings.sh_degree, shs, d) + 0.5
        clamped = raw < 0
        rgb = torch.clamp_min(raw, 0.0)

    return dict(depth=tz, radius=torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32),
                px=px, py=py, conic=torch.stack([cA, cB, cC], dim=-1), opacity=opacities.reshape(-1), rgb=rgb,
                rect=torch.stack([rminx, rminy, rmaxx, rmaxy], dim=-1), tiles_touched=tiles_touched * visible,
                visible=visible, cov3D=cov3D, clamped=clamped)


def rasterize(means3D, opacities, settings: RasterSettings, shs=None, colors_precomp=None, scales=None,
              rotations=None, cov3D_precomp=None, means2D=None, return_aux=False):
    """Full forward (A.1-A.3).  Returns (color[3,H,W], radii[P]) like the reference operator."""
    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    dt = means3D.dtype
    W, H = settings.image_width, settings.image_height
    g = preprocess(means3D, opacities, settings, shs, colors_precomp, scales, rotations, cov3D_precomp, means2D)
    vis = g["visible"]
    idx = torch.nonzero(vis).reshape(-1)
    # front-to-back order: depth bits ascending (depth > 0.2 so float order == bit order), ties by index
    depth32 = g["depth"].detach().to(torch.float32)[idx]
    order = torch.argsort(depth32.contiguous().view(torch.int32), stable=True)
    idx = idx[order]
    n = idx.numel()
    bg = settings.bg.to(dt).reshape(3)
    if n == 0:
        color = bg[:, None, None].expand(3, H, W).clone() + 0 * means3D.sum()
        return (color, g["radius"]) if not return_aux else (color, g["radius"], dict(num_rendered=0))

    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    pixx = xs.reshape(-1).to(dt)
    pixy = ys.reshape(-1).to(dt)
    tix = (xs.reshape(-1) // TILE)
    tiy = (ys.reshape(-1) // TILE)
    rect = g["rect"][idx]
    in_rect = ((tix[:, None] >= rect[None, :, 0]) & (tix[:, None] < rect[None, :, 2])
               & (tiy[:, None] >= rect[None, :, 1]) & (tiy[:, None] < rect[None, :, 3]))  # [N,n]
    dx = g["px"][idx][None, :] - pixx[:, None]
    dy = g["py"][idx][None, :] - pixy[:, None]
    con = g["conic"][idx]
    power = -0.5 * (con[None, :, 0] * dx * dx + con[None, :, 2] * dy * dy) - con[None, :, 1] * dx * dy
    Gv = torch.exp(power)
    raw_alpha = g["opacity"][idx][None, :] * Gv
    alpha = raw_alpha + (torch.clamp(raw_alpha, max=0.99) - raw_alpha).detach()  # straight-through min(0.99, .)
    keep = in_rect & (power.detach() <= 0) & (alpha.detach() >= 1.0 / 255.0)
    alpha = torch.where(keep, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - alpha
    Tincl = torch.cumprod(one_m, dim=1)
    Texcl = torch.cat([torch.ones_like(Tincl[:, :1]), Tincl[:, :-1]], dim=1)
    alive = Tincl.detach() >= 1e-4  # the Gaussian that would push T below 1e-4 is NOT blended
    wgt = torch.where(alive & keep, alpha * Texcl, torch.zeros_like(alpha))
    final_T = torch.where(alive, one_m, torch.ones_like(one_m)).prod(dim=1)
    rgb = g["rgb"][idx]
    color = wgt @ rgb + final_T[:, None] * bg[None, :]
    color = color.t().reshape(3, H, W)
    if return_aux:
        aux = dict(num_rendered=int(g["tiles_touched"].sum()), final_T=final_T.reshape(H, W), geom=g,
                   n_blended=(alive & keep).sum(dim=1).reshape(H, W))
        return color, g["radius"], aux
    return color, g["radius"]
