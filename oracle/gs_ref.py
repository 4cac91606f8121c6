"""ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED.

ctypes binding + autograd wrapper for oracle/gs_ref.c (the tile-based C restatement of the
rasterizer the reference calls at gaussian_renderer/__init__.py:126-135).  Used as the checker in
tests/, in __graft_entry__.smoke() and as bench.py's cpu_baseline leg ("kind": "port").
Never imported by instantsplat_amd/.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libgsref.so")
    src = os.path.join(_HERE, "gs_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "clean", "all"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        for suf in ("f32", "f64"):
            getattr(_LIB, f"gsref_forward_{suf}").restype = ctypes.c_void_p
            getattr(_LIB, f"gsref_num_rendered_{suf}").restype = ctypes.c_int64
            getattr(_LIB, f"gsref_num_rendered_{suf}").argtypes = [ctypes.c_void_p]
            getattr(_LIB, f"gsref_free_{suf}").argtypes = [ctypes.c_void_p]
    return _LIB


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class RefContext:
    def __init__(self, handle, suf):
        self.handle, self.suf = handle, suf

    def __del__(self):
        try:
            if self.handle:
                getattr(lib(), f"gsref_free_{self.suf}")(ctypes.c_void_p(self.handle))
                self.handle = None
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass

    @property
    def num_rendered(self):
        return int(getattr(lib(), f"gsref_num_rendered_{self.suf}")(ctypes.c_void_p(self.handle)))

    def aux(self, W, H):
        dt = torch.float32 if self.suf == "f32" else torch.float64
        T = ((W + 15) // 16) * ((H + 15) // 16)
        final_T = torch.empty(H, W, dtype=dt)
        n_contrib = torch.empty(H, W, dtype=torch.int32)
        tile_start = torch.empty(T + 1, dtype=torch.int64)
        lst = torch.empty(max(self.num_rendered, 1), dtype=torch.int32)
        getattr(lib(), f"gsref_get_aux_{self.suf}")(ctypes.c_void_p(self.handle), _ptr(final_T), _ptr(n_contrib),
                                                    _ptr(tile_start), _ptr(lst))
        return dict(final_T=final_T, n_contrib=n_contrib, tile_start=tile_start, list=lst[: self.num_rendered])


def forward(means3D, opacities, settings, shs=None, colors_precomp=None, scales=None, rotations=None,
            cov3D_precomp=None):
    """settings: any object with the 12 reference fields (oracle.raster_torch.RasterSettings)."""
    dt = means3D.dtype
    suf = "f32" if dt == torch.float32 else "f64"
    cr = ctypes.c_float if dt == torch.float32 else ctypes.c_double
    c = lambda t: None if t is None else t.detach().to(dt).contiguous()
    means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp = map(
        c, (means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp))
    view, proj, campos, bg = c(settings.viewmatrix.reshape(-1)), c(settings.projmatrix.reshape(-1)), c(settings.campos), c(settings.bg)
    P = means3D.shape[0]
    W, H = settings.image_width, settings.image_height
    M = shs.shape[1] if shs is not None else 0
    color = torch.empty(3, H, W, dtype=dt)
    radii = torch.zeros(P, dtype=torch.int32)
    fn = getattr(lib(), f"gsref_forward_{suf}")
    handle = fn(P, int(settings.sh_degree), M, W, H, _ptr(means3D), _ptr(shs), _ptr(colors_precomp), _ptr(opacities),
                _ptr(scales), cr(settings.scale_modifier), _ptr(rotations), _ptr(cov3D_precomp), _ptr(view), _ptr(proj),
                _ptr(campos), cr(settings.tanfovx), cr(settings.tanfovy), _ptr(bg), _ptr(color), _ptr(radii))
    ctx = RefContext(handle, suf)
    ctx.saved = (means3D, shs, scales, rotations, colors_precomp, cov3D_precomp, P, M)
    return color, radii, ctx


def backward(ctx: RefContext, dL_dpix):
    means3D, shs, scales, rotations, colors_precomp, cov3D_precomp, P, M = ctx.saved
    dt = means3D.dtype
    g = dict(means3D=torch.zeros(P, 3, dtype=dt), means2D=torch.zeros(P, 3, dtype=dt),
             shs=torch.zeros(P, max(M, 1), 3, dtype=dt) if shs is not None else None,
             colors=torch.zeros(P, 3, dtype=dt), opacities=torch.zeros(P, dtype=dt),
             scales=torch.zeros(P, 3, dtype=dt) if scales is not None else None,
             rotations=torch.zeros(P, 4, dtype=dt) if rotations is not None else None,
             cov3D=torch.zeros(P, 6, dtype=dt) if cov3D_precomp is not None else None)
    dL = dL_dpix.detach().to(dt).contiguous()
    getattr(lib(), f"gsref_backward_{ctx.suf}")(
        ctypes.c_void_p(ctx.handle), _ptr(means3D), _ptr(shs), _ptr(scales), _ptr(rotations), _ptr(dL),
        _ptr(g["means3D"]), _ptr(g["means2D"]), _ptr(g["shs"]), _ptr(g["colors"]), _ptr(g["opacities"]),
        _ptr(g["scales"]), _ptr(g["rotations"]), _ptr(g["cov3D"]))
    return g


class _RefRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, settings):
        e = lambda t: None if (t is None or t.numel() == 0) else t
        color, radii, rc = forward(means3D, opacities.reshape(-1), settings, e(shs), e(colors_precomp), e(scales),
                                   e(rotations), e(cov3D_precomp))
        ctx.rc = rc
        ctx.opac_shape = opacities.shape
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, g_color, _):
        g = backward(ctx.rc, g_color)
        none_if = lambda t, flag: t if flag else None
        has_sh = g["shs"] is not None
        return (g["means3D"], g["means2D"], g["shs"] if has_sh else None, None if has_sh else g["colors"],
                g["opacities"].reshape(ctx.opac_shape), g["scales"], g["rotations"], g["cov3D"], None)


def rasterize(means3D, means2D, opacities, settings, shs=None, colors_precomp=None, scales=None, rotations=None,
              cov3D_precomp=None):
    """Autograd-enabled CPU rasterizer with the reference operator's calling convention."""
    z = torch.empty(0)
    return _RefRasterize.apply(means3D, means2D, z if shs is None else shs, z if colors_precomp is None else colors_precomp,
                               opacities, z if scales is None else scales, z if rotations is None else rotations,
                               z if cov3D_precomp is None else cov3D_precomp, settings)
