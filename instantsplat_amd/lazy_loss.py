"""The reference's loss expression AS WRITTEN, on three launches instead of sixteen (reference train.py:171-176):

    Ll1 = l1_loss(image, gt_image)
    ssim_value = fused_ssim(image.unsqueeze(0), gt_image.unsqueeze(0))
    loss = (1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - ssim_value)
    loss.backward()

With the operator packages aliased and NO change to that source text (three launches):

  * `l1_loss(image, gt)` on an image that is being differentiated runs ONE pass over the two images
    (mi355gs_l1_ssim_pair_forward, one launch: d(ssim_mean)/d(image) and both means as per-workgroup partial sums) and keeps the result here, keyed on the two tensors'
    memory and version counters (and holding them, so the memory cannot be handed to another tensor meanwhile);
  * `fused_ssim(image[None], gt[None])` on the same two tensors takes the other half from that record: no launch;
  * both return a `LazyScalar` — a `torch.Tensor` subclass that RECORDS multiplication / division by Python numbers, addition
    and subtraction of Python numbers and of each other, and negation as a short postfix program instead of launching a
    kernel per operation (every such launch is a few microseconds the GPU idles through: the loop is launch-bound);
  * anything else that touches one — `.backward()`, `.item()`, `print`, any other torch function — first turns the recorded
    expression into an ordinary tensor: ONE autograd node on `image` whose forward finishes the two means and evaluates the program in
    one launch, with one float32 rounding per recorded operation (the bits eager PyTorch computes for the same expression), and whose backward is
    one launch over the image.

What cannot see the recorded expression — a C++ extension handed the object directly, `torch.autograd.backward(loss)` (which,
unlike `loss.backward()` and `torch.autograd.grad`, does not consult `__torch_function__`) — sees the subclass's own storage: a
NaN without a grad_fn, i.e. a loud failure, never a wrong number.  MI355GS_LAZY_LOSS=0 (or `lazy_loss.ENABLED = False`)
switches the whole mechanism off: `l1_loss` and `fused_ssim` are then two independent nodes again (loss_utils.py).
"""
from __future__ import annotations

import os
import time
import warnings

import numpy as np
import torch

from . import _lib

ENABLED = os.environ.get("MI355GS_LAZY_LOSS", "1") != "0"

# operation codes of include/mi355gs.h (MI355GS_LOSS_OP_*)
OP_L1, OP_SSIM, OP_MULK, OP_ADDK, OP_RSUBK, OP_DIVK, OP_NEG, OP_ADD, OP_SUB = range(9)
PROGRAM_MAX = 16

_PLACEHOLDER = {}


def _placeholder(dev):
    t = _PLACEHOLDER.get(dev)
    if t is None:
        t = _PLACEHOLDER[dev] = torch.full((), float("nan"), dtype=torch.float32, device=dev)
    return t


def _tensor_key(t):
    """what identifies "these bits": address, version counter, element count and the image dimensions"""
    return (t.data_ptr(), t._version, t.numel(), tuple(t.shape[-3:]))


class Pair:
    """One pair forward: the two tensors it was computed from (held: their memory stays theirs) and its results."""
    __slots__ = ("image", "gt", "image_key", "gt_key", "a", "b", "dmap", "means", "scratch", "dev")

    def __init__(self, image, gt):
        # means = [l1_mean, ssim_mean]: per-workgroup partial sums in `scratch` until the first materialisation finishes them
        ext = _lib.compiled()
        if ext is not None:
            self.means, self.scratch, self.dmap, self.a, self.b = ext.loss_pair_forward(image, gt)
        else:
            self.means, self.scratch, self.dmap, self.a, self.b = _pair_forward_ctypes(image, gt)
        self.image, self.gt = image, gt
        self.image_key, self.gt_key = _tensor_key(image), _tensor_key(gt)
        self.dev = self.a.device

    @property
    def l1(self):
        return self.means[0]

    @property
    def ssim(self):
        return self.means[1]

    def matches(self, img1, img2):
        return (_tensor_key(img1) == self.image_key and _tensor_key(img2) == self.gt_key and img1.is_contiguous() and img2.is_contiguous()
                and img1.dtype is torch.float32 and img2.dtype is torch.float32 and not img2.requires_grad)


_LAST = [None]   # the most recent pair (one training loop per process: reference utils/general_utils.py:133 pins one device)


def eligible(network_output, gt):
    """Is this l1_loss call the first half of the training loss?  An image that is being differentiated ([C,H,W] or
    [B,C,H,W], float32, contiguous) against data of the same shape on the same accelerator."""
    return (ENABLED and torch.is_grad_enabled() and network_output.requires_grad and not gt.requires_grad
            and network_output.dim() in (3, 4) and network_output.dtype is torch.float32 and gt.dtype is torch.float32
            and network_output.shape == gt.shape and network_output.numel() > 0 and network_output.device == gt.device
            and (network_output.is_cuda or _lib._TEST_MODE) and network_output.is_contiguous() and gt.is_contiguous()
            and network_output.shape[-3] * (network_output.shape[0] if network_output.dim() == 4 else 1) <= 65535)


def l1_of_pair(network_output, gt):
    """l1_loss(network_output, gt) as the first half of a pair: -> LazyScalar"""
    rec = _LAST[0] = Pair(network_output, gt)
    return _new(rec, ((OP_L1, 0.0),))


def ssim_of_pair(img1, img2):
    """fused_ssim(img1, img2) if the last l1_loss call was on the same two tensors: -> LazyScalar, else None"""
    rec = _LAST[0]
    if rec is None or not ENABLED or not torch.is_grad_enabled() or not rec.matches(img1, img2):
        return None
    return _new(rec, ((OP_SSIM, 0.0),))


def forget():
    """drop the record (and with it the two images and the gradient map it holds)"""
    _LAST[0] = None


# ---- the lazy scalar ---------------------------------------------------------------------------------------------------------
def _new(rec, prog):
    t = torch.Tensor._make_subclass(LazyScalar, _placeholder(rec.dev), False)
    t._rec, t._prog, t._real, t._slot = rec, prog, None, None
    return t


def _is_number(k):
    return isinstance(k, (int, float)) and not isinstance(k, bool)


def _is_lazy(x):
    return type(x) is LazyScalar


# ---- where `loss.item()` (train.py:188) finds its value: the program kernel stores (value, ticket) into a slot of pinned,
# device-mapped host memory as well, and item() polls the slot for its ticket — no device-to-host copy is enqueued and, above all,
# the read does not wait for the backward kernels queued behind the loss: the host goes on to optimizer.step() and the next
# frame's launches while the device still works on this iteration's backward (stream order keeps everything correct), so the
# device never idles through the reference loop's host work.  EARLY_ITEM = False restores the ordinary read (copy + stream wait).
EARLY_ITEM = os.environ.get("MI355GS_EARLY_ITEM", "1") != "0"
# loss.backward() of a recorded expression runs the autograd engine on the CALLING thread (no hand-off to the device's worker thread
# and back; csrc_torch/binding.cpp::loss_affine_backward).  MI355GS_BACKWARD_INLINE=0 leaves the engine's threading alone.
BACKWARD_INLINE = os.environ.get("MI355GS_BACKWARD_INLINE", "0") != "0"
_INLINE_SET = [None]
_NO_SLOT = torch.empty(0)   # "no host slot" for the compiled nodes (an empty tensor: pybind takes no None for a Tensor)
_SLOTS = {}          # device -> [pinned float32[N, 2], next index]
_N_SLOTS = 64
_TICKET = [0]


def _host_slot(dev):
    """-> (float32[2] view of a pinned slot, ticket): the slot is reused after _N_SLOTS materialisations, the ticket says whose value it holds"""
    ring = _SLOTS.get(dev)
    if ring is None:
        words = torch.zeros(_N_SLOTS, 2, dtype=torch.float32, pin_memory=(dev.type == "cuda"))
        ring = _SLOTS[dev] = [words, [words[i] for i in range(_N_SLOTS)], 0]
    k = ring[2]
    ring[2] = (k + 1) % _N_SLOTS
    t = _TICKET[0] = _TICKET[0] % 16000000 + 1    # exactly representable in float32, never 0
    return ring[1][k], float(t)


_PARTIALS = {}


def partials(prog):
    """(d value / d l1_mean, d value / d ssim_mean) of a recorded program, in float32 arithmetic: the numbers autograd's own
    backward of the eager expression multiplies the incoming gradient with, one rounding per operation."""
    c = _PARTIALS.get(prog)
    if c is not None:
        return c
    f = np.float32
    st = []
    for op, k in prog:
        k = f(k)
        if op == OP_L1:
            st.append((f(1), f(0)))
        elif op == OP_SSIM:
            st.append((f(0), f(1)))
        elif op == OP_MULK:
            a = st[-1]
            st[-1] = (a[0] * k, a[1] * k)
        elif op == OP_DIVK:
            a, r = st[-1], f(1) / k
            st[-1] = (a[0] * r, a[1] * r)
        elif op in (OP_RSUBK, OP_NEG):
            a = st[-1]
            st[-1] = (-a[0], -a[1])
        elif op == OP_ADD:
            b, a = st.pop(), st.pop()
            st.append((a[0] + b[0], a[1] + b[1]))
        elif op == OP_SUB:
            b, a = st.pop(), st.pop()
            st.append((a[0] - b[0], a[1] - b[1]))
        elif op != OP_ADDK:
            raise ValueError(f"unknown loss-program operation {op}")
    assert len(st) == 1, prog
    if len(_PARTIALS) > 256:
        _PARTIALS.clear()
    c = _PARTIALS[prog] = (float(st[0][0]), float(st[0][1]))
    return c


def materialize(x):
    """the recorded expression as an ordinary tensor: one node on the image (see the module docstring)"""
    rec = x._rec
    need_grad = torch.is_grad_enabled() and rec.image.requires_grad
    real = x._real
    if real is not None and (real.requires_grad or not need_grad):
        return real
    c_l1, c_ssim = partials(x._prog)
    ops, consts = [p[0] for p in x._prog], [float(p[1]) for p in x._prog]
    ext = _lib.compiled()
    if ext is not None:
        slot, ticket = _host_slot(rec.dev) if EARLY_ITEM else (None, 0.0)
        real = ext.loss_affine(rec.image, rec.a, rec.b, rec.dmap, rec.means, rec.scratch, ops, consts, c_l1, c_ssim, _NO_SLOT if slot is None else slot, ticket)
        x._slot = (slot, ticket) if slot is not None else None
    else:
        real = _LossAffine.apply(rec.image, rec, ops, consts, c_l1, c_ssim)
        x._slot = None
    x._real = real
    return real


def _unary(x, op, k=0.0):
    if len(x._prog) >= PROGRAM_MAX:
        return NotImplemented
    return _new(x._rec, x._prog + ((op, float(k)),))


def _binary(a, b, op):
    if a._rec is not b._rec or len(a._prog) + len(b._prog) >= PROGRAM_MAX:
        return NotImplemented
    return _new(a._rec, a._prog + b._prog + ((op, 0.0),))


def _h_mul(a, b):
    if _is_lazy(a) and _is_number(b):
        return _unary(a, OP_MULK, b)
    if _is_lazy(b) and _is_number(a):
        return _unary(b, OP_MULK, a)
    return NotImplemented


def _h_add(a, b, alpha=1):
    if alpha != 1:
        return NotImplemented
    if _is_lazy(a) and _is_lazy(b):
        return _binary(a, b, OP_ADD)
    if _is_lazy(a) and _is_number(b):
        return _unary(a, OP_ADDK, b)
    if _is_lazy(b) and _is_number(a):
        return _unary(b, OP_ADDK, a)
    return NotImplemented


def _h_sub(a, b, alpha=1):
    if alpha != 1:
        return NotImplemented
    if _is_lazy(a) and _is_lazy(b):
        return _binary(a, b, OP_SUB)
    if _is_lazy(a) and _is_number(b):
        return _unary(a, OP_ADDK, -b)     # x - k == x + (-k) in IEEE arithmetic
    if _is_lazy(b) and _is_number(a):
        return _unary(b, OP_RSUBK, a)
    return NotImplemented


def _h_rsub(a, b, alpha=1):   # rsub(a, b) = b - a
    return _h_sub(b, a, alpha)


def _h_div(a, b, rounding_mode=None):
    if rounding_mode is None and _is_lazy(a) and _is_number(b) and b != 0:
        return _unary(a, OP_DIVK, b)
    return NotImplemented


def _h_neg(a):
    return _unary(a, OP_NEG) if _is_lazy(a) else NotImplemented


_T = torch.Tensor
_HANDLERS = {}
for _fns, _h in (((_T.__mul__, _T.__rmul__, _T.mul, torch.mul, _T.multiply, torch.multiply), _h_mul),
                 ((_T.__add__, _T.__radd__, _T.add, torch.add), _h_add),
                 ((_T.__sub__, _T.sub, torch.sub, _T.subtract, torch.subtract), _h_sub),
                 ((_T.__rsub__, torch.rsub), _h_rsub),
                 ((_T.__truediv__, _T.div, torch.div, _T.true_divide, torch.true_divide, _T.divide, torch.divide), _h_div),
                 ((_T.__neg__, _T.neg, torch.neg, _T.negative, torch.negative), _h_neg)):
    for _f in _fns:
        _HANDLERS[_f] = _h


def _real_args(x):
    if _is_lazy(x):
        return materialize(x)
    if isinstance(x, (list, tuple)):
        return type(x)(_real_args(v) for v in x)
    if isinstance(x, dict):
        return {k: _real_args(v) for k, v in x.items()}
    return x


class LazyScalar(torch.Tensor):
    """A 0-dim float32 tensor whose value is a recorded scalar expression over the two means of one `Pair` (module docstring).

    The operators of train.py:176 and the two methods train.py:177,188 call are plain Python methods here — the interpreter
    reaches them without the `__torch_function__` protocol (about 1.5 us each instead of 4-5: the loop is bound by host time in
    exactly this stretch) —; everything else goes through `__torch_function__`, which materialises first."""

    def __mul__(self, k):
        return _unary(self, OP_MULK, k) if _is_number(k) and len(self._prog) < PROGRAM_MAX else _T.__mul__(self, k)

    __rmul__ = __mul__

    def __add__(self, k):
        if type(k) is LazyScalar:
            out = _binary(self, k, OP_ADD)
            return out if out is not NotImplemented else _T.__add__(self, k)
        return _unary(self, OP_ADDK, k) if _is_number(k) and len(self._prog) < PROGRAM_MAX else _T.__add__(self, k)

    __radd__ = __add__

    def __sub__(self, k):
        if type(k) is LazyScalar:
            out = _binary(self, k, OP_SUB)
            return out if out is not NotImplemented else _T.__sub__(self, k)
        return _unary(self, OP_ADDK, -k) if _is_number(k) and len(self._prog) < PROGRAM_MAX else _T.__sub__(self, k)

    def __rsub__(self, k):
        return _unary(self, OP_RSUBK, k) if _is_number(k) and len(self._prog) < PROGRAM_MAX else _T.__rsub__(self, k)

    def __truediv__(self, k):
        return _unary(self, OP_DIVK, k) if _is_number(k) and k != 0 and len(self._prog) < PROGRAM_MAX else _T.__truediv__(self, k)

    def __neg__(self):
        return _unary(self, OP_NEG) if len(self._prog) < PROGRAM_MAX else _T.__neg__(self)

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        """train.py:177.  The plain call — no explicit gradient, no graph to keep — materialises the expression and runs the
        engine in ONE call of the compiled binding, seeded with a cached 1 (no fill launch)."""
        if gradient is None and not retain_graph and not create_graph and inputs is None and torch.is_grad_enabled():
            real = self._real
            if real is None or not real.requires_grad:
                ext = _lib.compiled()
                rec = self._rec
                if ext is not None and rec.image.requires_grad:
                    if BACKWARD_INLINE is not _INLINE_SET[0]:
                        ext.backward_inline(bool(BACKWARD_INLINE))
                        _INLINE_SET[0] = BACKWARD_INLINE
                    c_l1, c_ssim = partials(self._prog)
                    prog = self._prog
                    slot, ticket = _host_slot(rec.dev) if EARLY_ITEM else (None, 0.0)
                    self._real = ext.loss_affine_backward(rec.image, rec.a, rec.b, rec.dmap, rec.means, rec.scratch, [p[0] for p in prog],
                                                          [p[1] for p in prog], c_l1, c_ssim, _NO_SLOT if slot is None else slot, ticket)
                    self._slot = (slot, ticket) if slot is not None else None
                    return None
        return materialize(self).backward(gradient, retain_graph, create_graph, inputs)

    def item(self):
        """train.py:188.  The value comes from the pinned slot the program kernel stored it in, if this materialisation has one
        (module comment at EARLY_ITEM): the same bits as the tensor's, without a copy and without waiting for the kernels
        enqueued behind the loss."""
        real = materialize(self)
        slot = self._slot
        if slot is not None:
            ext = _lib.compiled()
            if ext is not None:
                t0 = time.perf_counter()
                v = ext.wait_for_loss(slot[0], slot[1], 200_000)
                if v is not None:
                    return v
                if time.perf_counter() - t0 > 0.1:
                    # the slot never showed the value although the kernel ran (the ordinary read below returns it): host memory the
                    # device's stores do not reach coherently on this system.  Every later item() would spin for the whole timeout
                    # before falling back — switch the early read off for the process and say so once.
                    global EARLY_ITEM
                    EARLY_ITEM = False
                    warnings.warn("instantsplat_amd.lazy_loss: the pinned loss slot did not receive the value within 200 ms; "
                                  "loss.item() falls back to an ordinary read for the rest of the process (MI355GS_EARLY_ITEM=0)")
        return real.item()

    def __float__(self):
        return float(materialize(self).detach())

    def detach(self):
        return materialize(self).detach()

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        h = _HANDLERS.get(func)
        if h is not None:
            try:
                out = h(*args, **kwargs)
            except TypeError:   # an argument form the recorder does not know (out=, a third positional): the general path
                out = NotImplemented
            if out is not NotImplemented:
                return out
        args, kwargs = _real_args(args), _real_args(kwargs)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


# ---- ctypes twins of csrc_torch/binding.cpp::loss_pair_forward / LossAffineFn ---------------------------------------------------
def _pair_forward_ctypes(img1, img2):
    L = _lib.lib()
    (a, b), dev = _lib.f32c_on_one_device(img1, img2)
    if a.dim() not in (3, 4) or a.shape != b.shape or a.numel() == 0:
        raise RuntimeError("the loss pair expects two [C,H,W] or [B,C,H,W] tensors of equal shape")
    B = a.shape[0] if a.dim() == 4 else 1
    C, H, W = a.shape[-3:]
    scratch = torch.empty(int(L.mi355gs_ssim_scratch_bytes(B, C, H, W)), dtype=torch.uint8, device=dev)
    means = torch.empty(2, dtype=torch.float32, device=dev)
    dmap = torch.empty_like(a)
    with _lib.on_device(dev):
        _lib.check(L.mi355gs_l1_ssim_pair_forward(_lib.stream_ptr(dev), B, C, H, W, _lib.ptr(a), _lib.ptr(b), _lib.ptr(scratch),
                                                  _lib.ptr(dmap)), "l1_ssim_pair_forward")
    return means, scratch, dmap, a, b


class _LossAffine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, rec, ops, consts, c_l1, c_ssim):
        import ctypes
        L = _lib.lib()
        dev = rec.dev
        out = torch.empty((), dtype=torch.float32, device=dev)
        n = len(ops)
        a = rec.a
        B = a.shape[0] if a.dim() == 4 else 1
        C, H, W = a.shape[-3:]
        with _lib.on_device(dev):
            _lib.check(L.mi355gs_loss_program_eval(_lib.stream_ptr(dev), n, (ctypes.c_int32 * n)(*ops), (ctypes.c_float * n)(*consts), B, C, H, W,
                                                   _lib.ptr(rec.scratch), rec.means.data_ptr() + 4, rec.means.data_ptr(), _lib.ptr(out), None, 0.0),
                       "loss_program_eval")
        ctx.save_for_backward(rec.a, rec.b, rec.dmap)
        ctx.c, ctx.shape = (c_l1, c_ssim), image.shape
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, dmap = ctx.saved_tensors
        L = _lib.lib()
        g = _lib.f32c(g.reshape(1))
        d = torch.empty_like(a)
        with _lib.on_device(a.device):
            _lib.check(L.mi355gs_l1_ssim_pair_backward(_lib.stream_ptr(a.device), a.numel(), _lib.ptr(a), _lib.ptr(b), _lib.ptr(dmap),
                                                       _lib.ptr(g), ctx.c[0], _lib.ptr(g), ctx.c[1], _lib.ptr(d)), "l1_ssim_pair_backward")
        return d.view(ctx.shape), None, None, None, None, None
