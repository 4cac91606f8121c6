"""Shared helpers: run the same seeded scene through the oracle and through instantsplat_amd."""
import math

import torch

from instantsplat_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from instantsplat_amd.synthetic import syn_blob
from oracle import gs_ref
from oracle import raster_torch as rt


def settings_for(cam, deg, cls, bg, device="cpu", mod=1.0, debug=False):
    dev = torch.device(device)
    return cls(cam.image_height, cam.image_width, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), bg.to(dev), mod,
               torch.eye(4, device=dev), cam.projection_matrix.to(dev), deg, torch.zeros(3, device=dev), False, debug)


def relerr(a, b):
    return float((a.double().cpu() - b.double().cpu()).norm() / (b.double().cpu().norm() + 1e-30))


def run_blob_case(device, P, W, H, deg, scale_mean=0.05, seed=1, opacity="random", bg=(0.2, 0.5, 0.9), mod=1.0,
                  precomp_color=False, precomp_cov=False, backward=True, scale_grad_exact=False, with_f64=False):
    """Returns dict(ref=..., dut=...) each with color, radii, grads (dict); with_f64 adds "f64": the same case through the
    float64 build of the oracle (the yardstick for assert_no_worse_than_fp32_oracle).  scale_grad_exact: both sides return the true
    derivative with respect to `scales` instead of the published operator's dL/d(mod * scale) (include/mi355gs.h,
    mi355gs_tune_scale_grad; oracle/gs_ref.c header) — the two only differ at mod != 1."""
    from instantsplat_amd import _lib
    old = (gs_ref.lib().gsref_set_scale_grad_exact(int(scale_grad_exact)), _lib.lib().mi355gs_tune_scale_grad(int(scale_grad_exact)))
    try:
        return _run_blob_case(device, P, W, H, deg, scale_mean, seed, opacity, bg, mod, precomp_color, precomp_cov, backward, with_f64)
    finally:
        gs_ref.lib().gsref_set_scale_grad_exact(old[0])
        _lib.lib().mi355gs_tune_scale_grad(old[1])


def _run_blob_case(device, P, W, H, deg, scale_mean, seed, opacity, bg, mod, precomp_color, precomp_cov, backward, with_f64=False):
    sc = syn_blob(P, W, H, seed=seed, scale_mean=scale_mean, opacity=opacity)
    bg_t = torch.tensor(bg, dtype=torch.float32)
    torch.manual_seed(seed + 100)
    wgt = torch.randn(3, H, W)
    out = {}
    for which in ("ref", "dut") + (("f64",) if with_f64 else ()):
        dev = torch.device(device) if which == "dut" else torch.device("cpu")
        dt = torch.float64 if which == "f64" else torch.float32
        leaves = dict(means3D=sc.means3D.clone(), scaling=sc.scaling_logit.clone(), rot=sc.rotation.clone(),
                      op=sc.opacity_logit.clone(), shs=sc.shs.clone())
        leaves = {k: v.to(dev, dt).requires_grad_(True) for k, v in leaves.items()}
        m2d = torch.zeros(P, 3, device=dev, dtype=dt, requires_grad=True)
        kw = {}
        colors = cov = None
        if precomp_color:
            colors = torch.sigmoid(leaves["shs"][:, 0, :])
            kw["colors_precomp"] = colors
        else:
            kw["shs"] = leaves["shs"]
        if precomp_cov:
            cov = rt.cov3d_from_scale_rot(torch.exp(leaves["scaling"]), mod, leaves["rot"])
            kw["cov3D_precomp"] = cov
        else:
            kw["scales"] = torch.exp(leaves["scaling"])
            kw["rotations"] = leaves["rot"]
        if which != "dut":
            st = settings_for(sc.camera, deg, rt.RasterSettings, bg_t, mod=mod)
            if which == "f64":
                st = rt.RasterSettings(*[(x.double() if isinstance(x, torch.Tensor) else x) for x in st])
            color, radii = gs_ref.rasterize(leaves["means3D"], m2d, torch.sigmoid(leaves["op"]), st, **kw)
        else:
            st = settings_for(sc.camera, deg, GaussianRasterizationSettings, bg_t, device=dev, mod=mod)
            color, radii = GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=m2d, opacities=torch.sigmoid(leaves["op"]), **kw)
        grads = {}
        if backward:
            (color * wgt.to(dev, dt)).sum().backward()
            grads = {k: v.grad.detach().cpu().clone() for k, v in leaves.items() if v.grad is not None}
            grads["means2D"] = m2d.grad.detach().cpu().clone()
        out[which] = dict(color=color.detach().cpu(), radii=radii.cpu(), grads=grads)
    return out


def assert_raster_parity(out, fwd_tol=1e-4, fwd_max=5e-3, grad_tol=1e-4, radii_frac=1e-4):
    """Stated fp32 tolerance: >=99.99% of pixels within 1e-4, all within 5e-3 (isolated
    threshold flips), radii equal up to 0.01% off-by-one, per-tensor gradient relative L2 <= 1e-4."""
    ref, dut = out["ref"], out["dut"]
    d = (ref["color"] - dut["color"]).abs()
    frac_bad = float((d > fwd_tol).float().mean())
    assert frac_bad <= 1e-4, f"{frac_bad:.2e} of pixel values differ by more than {fwd_tol}"
    assert float(d.max()) <= fwd_max, f"max pixel error {float(d.max()):.3e}"
    mism = (ref["radii"] != dut["radii"])
    assert float(mism.float().mean()) <= radii_frac, f"{int(mism.sum())} radii differ"
    assert int((ref["radii"] - dut["radii"]).abs().max()) <= 1
    for k, g in ref["grads"].items():
        e = relerr(dut["grads"][k], g)
        assert e <= grad_tol, f"grad {k}: rel L2 {e:.3e} > {grad_tol}"


def run_custom_case(device, means, scales, rots, opac, colors, W, H, fovx_deg=60.0, bg=(0.1, 0.2, 0.3)):
    """Hand-built Gaussians (camera frame, precomputed colours) through oracle and device path; forward + backward."""
    from instantsplat_amd.camera import Camera
    tanx = math.tan(math.radians(fovx_deg) / 2)
    cam = Camera(0, torch.eye(4), math.radians(fovx_deg), 2 * math.atan(tanx * H / W), W, H)
    bg_t = torch.tensor(bg, dtype=torch.float32)
    P = means.shape[0]
    torch.manual_seed(3)
    wgt = torch.randn(3, H, W)
    out = {}
    for which in ("ref", "dut"):
        dev = torch.device("cpu") if which == "ref" else torch.device(device)
        lv = {k: v.clone().to(dev).requires_grad_(True) for k, v in dict(means3D=means, scales=scales, rot=rots, op=opac, col=colors).items()}
        m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
        if which == "ref":
            st = settings_for(cam, 0, rt.RasterSettings, bg_t)
            color, radii = gs_ref.rasterize(lv["means3D"], m2d, lv["op"], st, colors_precomp=lv["col"], scales=lv["scales"], rotations=lv["rot"])
        else:
            st = settings_for(cam, 0, GaussianRasterizationSettings, bg_t, device=dev)
            color, radii = GaussianRasterizer(st)(means3D=lv["means3D"], means2D=m2d, opacities=lv["op"], colors_precomp=lv["col"],
                                                  scales=lv["scales"], rotations=lv["rot"])
        (color * wgt.to(dev)).sum().backward()
        grads = {k: v.grad.detach().cpu().clone() for k, v in lv.items()}
        grads["means2D"] = m2d.grad.detach().cpu().clone()
        out[which] = dict(color=color.detach().cpu(), radii=radii.cpu(), grads=grads)
    # float64 oracle: the yardstick when the case is ill-conditioned in fp32
    lv = {k: v.clone().double().requires_grad_(True) for k, v in dict(means3D=means, scales=scales, rot=rots, op=opac, col=colors).items()}
    m2d = torch.zeros(P, 3, dtype=torch.float64, requires_grad=True)
    st = settings_for(cam, 0, rt.RasterSettings, bg_t)
    st64 = rt.RasterSettings(*[(x.double() if isinstance(x, torch.Tensor) else x) for x in st])
    color, _ = gs_ref.rasterize(lv["means3D"], m2d, lv["op"], st64, colors_precomp=lv["col"], scales=lv["scales"], rotations=lv["rot"])
    (color * wgt.double()).sum().backward()
    g64 = {k: v.grad.detach().clone() for k, v in lv.items()}
    g64["means2D"] = m2d.grad.detach().clone()
    out["f64"] = dict(color=color.detach(), grads=g64)
    return out


def assert_no_worse_than_fp32_oracle(out, factor=2.0, floor=1e-4):
    """For ill-conditioned inputs both fp32 implementations drift from the float64 result; require the device path
    to be within `factor` x the fp32 oracle's own error (or `floor`) for every gradient tensor and for the image."""
    t = out["f64"]
    e_img_ref = float((out["ref"]["color"].double() - t["color"]).abs().max())
    e_img_dut = float((out["dut"]["color"].double() - t["color"]).abs().max())
    assert e_img_dut <= max(factor * e_img_ref, floor), (e_img_dut, e_img_ref)
    for k, g in t["grads"].items():
        n = float(g.norm()) + 1e-30
        e_ref = float((out["ref"]["grads"][k].double() - g).norm()) / n
        e_dut = float((out["dut"]["grads"][k].double() - g).norm()) / n
        assert e_dut <= max(factor * e_ref, floor), (k, e_dut, e_ref)
