"""PerPointAdam on the HIP path — same constructor, param-group keys and update rule as reference
scene/per_point_adam.py:4-100, one fused kernel per tensor (mi355gs_adam_step) instead of ~15 eager
elementwise launches.

Semantics kept on purpose:
  * the moment update is gated on the WHOLE tensor's gradient norm being > 0 (a 0-dim mask);
  * eps is added to sqrt(v) before the bias correction is applied (it is folded into step_size);
  * `per_point_lr` multiplies the step of every element of a point's row and stays constant (the
    reference computes an adjusted multiplier and drops it).
"""
from __future__ import annotations

import ctypes
import math

import torch
from torch.optim import Optimizer

from . import _lib


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear LR decay with optional delayed warm-up (semantics of reference utils/general_utils.py:29-62)."""

    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        if lr_delay_steps > 0:
            delay_rate = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
        else:
            delay_rate = 1.0
        t = min(max(step / max_steps, 0.0), 1.0)
        return delay_rate * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)

    return helper


class PerPointAdam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if not all(0.0 <= x for x in [lr, eps, weight_decay]):
            raise ValueError(f"Invalid learning parameters: lr={lr}, eps={eps}, weight_decay={weight_decay}")
        if not all(0.0 <= beta < 1.0 for beta in betas):
            raise ValueError(f"Invalid beta parameters: {betas}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, per_point_lr=None))

    @torch.no_grad()
    def step(self, closure=None):
        """All parameter tensors in two launches (mi355gs_adam_multi_step): per-tensor sum of squared gradients
        for the whole-tensor gate, then the fused update."""
        loss = closure() if closure is not None else None
        L = _lib.lib()
        items = []
        for group in self.param_groups:
            per_point_lr = group.get("per_point_lr")
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                grad = p.grad
                if grad.is_sparse:
                    raise RuntimeError("PerPointAdam does not support sparse gradients")
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p)
                    state["exp_avg_sq"] = torch.zeros_like(p)
                state["step"] += 1
                if group["weight_decay"] != 0:
                    grad = grad.add(p, alpha=group["weight_decay"])
                grad = _lib.f32c(grad)
                pplr, row = None, 1
                if per_point_lr is not None:
                    if not isinstance(per_point_lr, torch.Tensor):
                        raise TypeError("per_point_lr must be a torch.Tensor")
                    if per_point_lr.device != p.device:
                        raise ValueError("per_point_lr must be on the same device as parameter")
                    expected_shape = p.shape[:1] + (1,) * (p.dim() - 1)
                    if per_point_lr.shape != expected_shape:
                        raise ValueError(f"Invalid per_point_lr shape. Expected {expected_shape}, got {per_point_lr.shape}")
                    pplr = _lib.f32c(per_point_lr)
                    row = p.numel() // p.shape[0]
                items.append((p, grad, state, pplr, row, float(group["lr"]), float(beta1), float(beta2), float(group["eps"])))
        # one call per (betas, eps) combination, at most 8 tensors each
        while items:
            b1, b2, eps = items[0][6:9]
            batch = [it for it in items if it[6:9] == (b1, b2, eps)][:8]
            items = [it for it in items if not any(it is b for b in batch)]
            n = len(batch)
            dev = _lib.require_device(*[t for it in batch for t in (it[0], it[1], it[2]["exp_avg"], it[2]["exp_avg_sq"], it[3])])
            I64, I32, PTR, F32 = ctypes.c_int64 * n, ctypes.c_int32 * n, ctypes.c_void_p * n, ctypes.c_float * n
            addr = lambda t: 0 if t is None else t.data_ptr()
            scratch = torch.empty(8, dtype=torch.float32, device=dev)
            _lib.check(L.mi355gs_adam_multi_step(
                _lib.stream_ptr(dev), n, I64(*[it[0].numel() for it in batch]), I32(*[it[4] for it in batch]),
                PTR(*[addr(it[0]) for it in batch]), PTR(*[addr(it[1]) for it in batch]),
                PTR(*[addr(it[2]["exp_avg"]) for it in batch]), PTR(*[addr(it[2]["exp_avg_sq"]) for it in batch]),
                PTR(*[addr(it[3]) for it in batch]), F32(*[it[5] for it in batch]), b1, b2, eps,
                I32(*[it[2]["step"] for it in batch]), _lib.ptr(scratch)), "adam_multi_step")
        return loss
