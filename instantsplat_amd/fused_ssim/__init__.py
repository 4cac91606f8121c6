"""Drop-in for the `fused_ssim` package (reference train.py:39-43,173).

    fused_ssim(img1, img2, padding="same", train=True) -> 0-dim tensor, gradient w.r.t. img1 only.

`fused_l1_ssim_loss` additionally folds train.py:171-176 — (1-lambda)*L1 + lambda*(1-SSIM) — into
the same two kernels (the L1 term costs no extra pass over the images).
"""
from __future__ import annotations

import torch

from .. import _lib
from .. import lazy_loss


def _run_forward(img1, img2, train, valid=False):
    L = _lib.lib()
    a, b = _lib.f32c(img1), _lib.f32c(img2)
    if a.dim() != 4 or a.shape != b.shape:
        raise RuntimeError("fused_ssim expects two [B,C,H,W] tensors of equal shape")
    dev = _lib.require_device(a, b)
    B, C, H, W = a.shape
    new = lambda: torch.empty_like(a)
    dm1, dm2, dm3 = (new(), new(), new()) if train else (None, None, None)
    scratch = torch.empty(int(L.mi355gs_ssim_scratch_bytes(B, C, H, W)), dtype=torch.uint8, device=dev)
    if valid and (H <= 10 or W <= 10):
        raise RuntimeError('fused_ssim(padding="valid") needs images larger than the 11x11 window')
    out = torch.empty(2, dtype=torch.float32, device=dev)  # [ssim_mean, l1_mean]
    with _lib.on_device(dev):
        _lib.check(L.mi355gs_ssim_forward(_lib.stream_ptr(dev), B, C, H, W, _lib.ptr(a), _lib.ptr(b), _lib.ptr(dm1), _lib.ptr(dm2),
                                          _lib.ptr(dm3), _lib.ptr(scratch), _lib.ptr(out[0:1]), None if valid else _lib.ptr(out[1:2]),
                                          1 if valid else 0), "ssim_forward")
    return a, b, dm1, dm2, dm3, out


def _run_backward(a, b, dm1, dm2, dm3, ssim_scale, l1_scale, valid=False):
    L = _lib.lib()
    dev = a.device
    B, C, H, W = a.shape
    grad = torch.empty_like(a)
    with _lib.on_device(dev):
        _lib.check(L.mi355gs_ssim_backward(_lib.stream_ptr(dev), B, C, H, W, _lib.ptr(a), _lib.ptr(b), _lib.ptr(dm1), _lib.ptr(dm2),
                                           _lib.ptr(dm3), _lib.ptr(ssim_scale), _lib.ptr(l1_scale), _lib.ptr(grad), 1 if valid else 0),
                   "ssim_backward")
    return grad


class _FusedSSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2, train, valid=False):
        a, b, dm1, dm2, dm3, out = _run_forward(img1, img2, train, valid)
        if train:
            ctx.save_for_backward(a, b, dm1, dm2, dm3)
        ctx.train, ctx.valid = train, valid
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        if not ctx.train:
            raise RuntimeError("fused_ssim was called with train=False; no gradient is available")
        a, b, dm1, dm2, dm3 = ctx.saved_tensors
        scale = _lib.f32c(g.reshape(1))
        return _run_backward(a, b, dm1, dm2, dm3, scale, None, ctx.valid), None, None, None


def fused_ssim(img1, img2, padding="same", train=True):
    """padding="same" (the reference's use, train.py:173): zero padding, mean over the whole map.  padding="valid": the
    mean (and the gradient) only covers the region where the 11x11 window lies inside the image."""
    if padding not in ("same", "valid"):
        raise ValueError(f'padding must be "same" or "valid", got {padding!r}')
    if train and padding == "same" and lazy_loss._LAST[0] is not None and isinstance(img1, torch.Tensor) and isinstance(img2, torch.Tensor):
        # the second half of train.py:171-176: `l1_loss(image, gt)` has just computed SSIM of these two tensors as well (lazy_loss.py)
        half = lazy_loss.ssim_of_pair(img1, img2)
        if half is not None:
            return half
    ext = _lib.compiled()
    if ext is not None:   # the same two C-ABI calls from a C++ autograd node (csrc_torch/binding.cpp::SsimFn)
        return ext.fused_ssim(img1, img2, bool(train), padding == "valid")
    return _FusedSSIM.apply(img1, img2, train, padding == "valid")


_SCRATCH_BYTES = {}


def _scratch_bytes(L, B, C, H, W):
    key = (B, C, H, W)
    n = _SCRATCH_BYTES.get(key)
    if n is None:
        if len(_SCRATCH_BYTES) > 64:
            _SCRATCH_BYTES.clear()
        n = _SCRATCH_BYTES[key] = int(L.mi355gs_ssim_scratch_bytes(B, C, H, W))
    return n


class _FusedL1SSIM(torch.autograd.Function):
    """(1-lambda)*L1 + lambda*(1-SSIM): loss AND its gradient w.r.t. img1 come out of one pass over the images
    (mi355gs_l1_ssim_loss_fused); backward only scales the kept gradient by the incoming dL/dloss."""

    @staticmethod
    def forward(ctx, img1, img2, lambda_dssim):
        L = _lib.lib()
        (a, b), dev = _lib.f32c_on_one_device(img1, img2)
        if a.dim() != 4 or a.shape != b.shape:
            raise RuntimeError("fused_ssim expects two [B,C,H,W] tensors of equal shape")
        B, C, H, W = a.shape
        scratch = torch.empty(_scratch_bytes(L, B, C, H, W), dtype=torch.uint8, device=dev)
        out = torch.empty(2, dtype=torch.float32, device=dev)   # [ssim_mean, l1_mean]
        loss = torch.empty((), dtype=torch.float32, device=dev)
        grad = torch.empty_like(a)
        p0 = out.data_ptr()
        with _lib.on_device(dev):
            _lib.check(L.mi355gs_l1_ssim_loss_fused(_lib.stream_ptr(dev), B, C, H, W, _lib.ptr(a), _lib.ptr(b), _lib.ptr(scratch),
                                                    float(lambda_dssim), p0, p0 + 4, _lib.ptr(loss), _lib.ptr(grad)),
                       "l1_ssim_loss_fused")
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(out)
        return loss, out

    @staticmethod
    def backward(ctx, g, _):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def fused_l1_ssim_loss(img1, img2, lambda_dssim=0.2):
    """Returns (loss, [ssim_mean, l1_mean]); loss = (1-lambda)*L1 + lambda*(1-SSIM) as reference train.py:176."""
    ext = _lib.compiled()
    if ext is not None:
        return ext.l1_ssim_loss(img1, img2, float(lambda_dssim))
    return _FusedL1SSIM.apply(img1, img2, lambda_dssim)
