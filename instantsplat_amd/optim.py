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
from torch.optim.optimizer import _global_optimizer_post_hooks as _global_post_hooks
from torch.optim.optimizer import _global_optimizer_pre_hooks as _global_pre_hooks

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
    # Contract of the gate shortcut (compiled binding only): a `.grad` that is exactly the tensor the last render backward
    # returned — same storage, offset 0, same size, unchanged `_version()` — takes the whole-tensor gate flag that backward left
    # on the device instead of a pass over the gradient.  `_version()` sees every in-place torch op on the tensor itself; it does
    # NOT see writes through `.grad.data` / `.detach()` views made before the backward, or raw-pointer kernels.  Code that edits
    # gradients that way sets `use_backward_gates = False` (class or instance): every step then sums the squared gradients itself,
    # exactly like the reference's `grad.norm() > 0` (one more ~20 us launch per step), and the render backward goes back to a
    # fresh zero tensor for `f_rest`'s gradient below its SH degree instead of aliases of one persistent zero buffer
    # (csrc_torch/binding.cpp::zero_grad_like: same contract — in-place torch ops on such a `.grad` are seen and handled).
    use_backward_gates = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if not all(0.0 <= x for x in [lr, eps, weight_decay]):
            raise ValueError(f"Invalid learning parameters: lr={lr}, eps={eps}, weight_decay={weight_decay}")
        if not all(0.0 <= beta < 1.0 for beta in betas):
            raise ValueError(f"Invalid beta parameters: {betas}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, per_point_lr=None))
        # The kernels update parameters in place through raw pointers, row-major.  The reference's GaussianModel hands over a
        # COLUMN-major `_xyz` whenever the points come from a PLY file (`np.vstack([x, y, z]).T` through `torch.tensor`,
        # scene/dataset_readers.py:217 -> scene/gaussian_model.py:148, keeps the strides): re-lay such a parameter out once, here —
        # same Parameter object, same values, and everything else that touches it (elementwise torch code, the operators'
        # own `.contiguous()`) is layout-agnostic.
        with torch.no_grad():
            for group in self.param_groups:
                for p in group["params"]:
                    if not p.is_contiguous():
                        p.data = p.data.contiguous()

    def zero_grad(self, set_to_none: bool = True):
        """Same contract as torch.optim.Optimizer.zero_grad (the reference calls it with set_to_none=True, train.py:211),
        without the per-call profiler scopes and hook dispatch of the generic implementation (~25 us per iteration)."""
        fast = self.__dict__.get("_fast")
        if set_to_none and fast is not None and fast[4] and fast[3] == sum(len(g["params"]) for g in self.param_groups):
            fast[2].zero_owned()   # every parameter of the optimizer is one of the compiled plan's: cleared from C++
            return
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.detach_()
                        p.grad.requires_grad_(False)
                        p.grad.zero_()

    @staticmethod
    def _group_sig(group):
        pp = group.get("per_point_lr")
        return (group["betas"], group["eps"], pp, None if pp is None else pp.data_ptr())

    @staticmethod
    def _same_sig(a, b):
        return a[0] == b[0] and a[1] == b[1] and a[2] is b[2] and a[3] == b[3]

    def load_state_dict(self, state_dict):
        self._plans = {}   # the moments are new tensors
        self._fast = None
        return super().load_state_dict(state_dict)

    def _plan_for(self, live):
        """Everything about a step that does not change from one iteration to the next — which tensors take part, their
        sizes, the addresses of parameters / moments / per-point multipliers, the (betas, eps) batches — as ready-made ctypes
        arrays, keyed by the identity of the participating parameters and checked against their current addresses, moment
        tensors and group settings."""
        key = tuple([id(p) for _, p in live])
        plans = self.__dict__.get("_plans")
        plan = plans.get(key) if plans is not None else None
        if plan is not None:
            state_of, same = self.state, self._same_sig
            for (g, p), (a, m, sig) in zip(live, plan["check"]):
                if p.data_ptr() != a or state_of[p]["exp_avg"] is not m or not same(self._group_sig(g), sig):
                    break
            else:
                return plan
        if not hasattr(self, "_plans"):
            self._plans = {}
        items = []
        for group, p in live:
            per_point_lr = group.get("per_point_lr")
            beta1, beta2 = group["betas"]
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
            items.append((group, p, pplr, row, (float(beta1), float(beta2), float(group["eps"]))))
        batches = []
        while items:   # one call per (betas, eps) combination, at most 8 tensors each
            hyper = items[0][4]
            batch = [it for it in items if it[4] == hyper][:8]
            items = [it for it in items if not any(it is b for b in batch)]
            n = len(batch)
            st = [self.state[it[1]] for it in batch]
            dev = _lib.require_device(*[t for it, s_ in zip(batch, st) for t in (it[1], s_["exp_avg"], s_["exp_avg_sq"], it[2])])
            I64, I32, PTR = ctypes.c_int64 * n, ctypes.c_int32 * n, ctypes.c_void_p * n
            batches.append(dict(
                n=n, dev=dev, hyper=hyper, groups=[it[0] for it in batch], params=[it[1] for it in batch], states=st,
                numel=I64(*[it[1].numel() for it in batch]), row=I32(*[it[3] for it in batch]),
                p=PTR(*[it[1].data_ptr() for it in batch]), m=PTR(*[s_["exp_avg"].data_ptr() for s_ in st]),
                v=PTR(*[s_["exp_avg_sq"].data_ptr() for s_ in st]), pplr=PTR(*[(0 if it[2] is None else it[2].data_ptr()) for it in batch]),
                keep=[it[2] for it in batch], PTR=PTR, F32=ctypes.c_float * n, I32=I32))
        plan = dict(batches=batches, check=[(p.data_ptr(), self.state[p]["exp_avg"], self._group_sig(g)) for g, p in live])
        self._plans[key] = plan
        return plan

    def step(self, closure=None):
        """All parameter tensors in two launches (mi355gs_adam_multi_step): per-tensor sum of squared gradients
        for the whole-tensor gate, then the fused update.

        torch.optim.Optimizer would wrap this method in its profiler scope + hook dispatcher (~20 us of host time per call, as
        much as the rest of the step); the method is marked `hooked` so the base class leaves it alone, and does both itself
        — only when a profiler is running or a hook is registered."""
        if _global_pre_hooks or _global_post_hooks or self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks \
                or torch.autograd._profiler_enabled():
            return self._step_with_hooks(closure)
        return self._step(closure)

    step.hooked = True   # see torch.optim.Optimizer._patch_step_function

    def _step_with_hooks(self, closure):
        args, kwargs = (self, closure), {}
        with torch.autograd.profiler.record_function(f"Optimizer.step#{self.__class__.__name__}.step"):
            for hook in (*_global_pre_hooks.values(), *self._optimizer_step_pre_hooks.values()):
                result = hook(self, args, kwargs)
                if result is not None:
                    if not (isinstance(result, tuple) and len(result) == 2):
                        raise RuntimeError(f"{hook} must return None or a tuple of (new_args, new_kwargs), but got {result}.")
                    args, kwargs = result
            out = self._step(*args[1:], **kwargs)
            self._optimizer_step_code()
            for hook in (*self._optimizer_step_post_hooks.values(), *_global_post_hooks.values()):
                hook(self, args, kwargs)
            return out

    def _step_fast(self, fast):
        """The steady state of a training loop — the same tensors with a gradient each as in the previous step, one (betas,
        eps) batch, no weight decay — without rebuilding anything: validate, one call into the compiled plan (which reads the
        gradients off the Parameters itself), bump the step counts.  Returns False (having changed nothing) when anything
        differs; the general path then takes the step."""
        entries, plan = fast[1], fast[2]
        lrs, steps = [], []
        for group, p, st, ptr, sig, m in entries:
            pp = group.get("per_point_lr")
            if st.get("exp_avg") is not m or group["weight_decay"] != 0 or group["betas"] != sig[0] or group["eps"] != sig[1] \
                    or pp is not sig[2] or (pp is not None and pp.data_ptr() != sig[3]):
                return False
            lrs.append(group["lr"])
            steps.append(st["step"] + 1)
        if not self.use_backward_gates:
            fast[0].forget_gates()
            fast[0].shared_zero_grad(False)   # gradients are edited behind the version counter: no shared zero buffer either
        if not plan.step_owned(lrs, steps):   # a parameter without a (dense) gradient, or re-allocated: nothing was done
            return False
        for (_, _, st, _, _, _), k in zip(entries, steps):
            st["step"] = k
        return True

    def _step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ext = _lib.compiled()
        fast = self.__dict__.get("_fast")
        if fast is not None:
            if fast[0] is ext and fast[3] == sum(len(g["params"]) for g in self.param_groups) and self._step_fast(fast):
                return loss
            self._fast = None
        state_of = self.state
        live = []
        for group in self.param_groups:
            for p in group["params"]:
                grad = p.grad
                if grad is None:
                    continue
                if grad.is_sparse:
                    raise RuntimeError("PerPointAdam does not support sparse gradients")
                state = state_of[p]
                if len(state) == 0:
                    state["step"] = 0
                    with torch.no_grad():
                        state["exp_avg"] = torch.zeros_like(p)
                        state["exp_avg_sq"] = torch.zeros_like(p)
                state["step"] += 1
                live.append((group, p))
        if not live:
            return loss
        L = _lib.lib()
        f32 = torch.float32
        batches = self._plan_for(live)["batches"]
        for b in batches:
            dev = b["dev"]
            if ext is not None and not any(group["weight_decay"] != 0 for group in b["groups"]):
                # compiled binding: the same library call from C++ (csrc_torch/binding.cpp AdamPlan), which also recognises
                # gradients that are exactly the tensors the last render backward wrote and then takes that call's gate flags
                # instead of launching the pass over all gradients
                plan = b.get("compiled")
                if plan is None or b.get("compiled_ext") is not ext:
                    b1, b2, eps = b["hyper"]
                    plan = b["compiled"] = ext.AdamPlan([p.data for p in b["params"]], [s_["exp_avg"] for s_ in b["states"]],
                                                        [s_["exp_avg_sq"] for s_ in b["states"]], b["keep"], b1, b2, eps)
                    b["compiled_ext"] = ext
                if not self.use_backward_gates:
                    ext.forget_gates()
                    ext.shared_zero_grad(False)
                plan.step([p.grad for p in b["params"]], [group["lr"] for group in b["groups"]], [s_["step"] for s_ in b["states"]])
                if len(batches) == 1 and len(live) == sum(len(g["params"]) for g in self.param_groups):
                    # every parameter of the optimizer took part, in one batch: remember the line-up for _step_fast
                    plan.set_owners(list(b["params"]))
                    self._fast = (ext, [(g_, p_, s_, p_.data_ptr(), self._group_sig(g_), s_["exp_avg"]) for g_, p_, s_ in zip(b["groups"], b["params"], b["states"])],
                                  plan, len(live), True)
                continue
            ptrs = []
            keep = []
            for group, p in zip(b["groups"], b["params"]):
                g = p.grad
                if group["weight_decay"] != 0:
                    with torch.no_grad():
                        g = g.add(p, alpha=group["weight_decay"])
                if g.dtype is not f32 or not g.is_contiguous():
                    g = _lib.f32c(g)
                if g.device != dev:
                    raise RuntimeError(f"tensors on different devices: {dev} vs {g.device}")
                keep.append(g)
                ptrs.append(g.data_ptr())
            scratch = torch.empty(8, dtype=f32, device=dev)
            b1, b2, eps = b["hyper"]
            with _lib.on_device(dev):
                _lib.check(L.mi355gs_adam_multi_step(
                    _lib.stream_ptr(dev), b["n"], b["numel"], b["row"], b["p"], b["PTR"](*ptrs), b["m"], b["v"],
                    b["pplr"], b["F32"](*[group["lr"] for group in b["groups"]]), b1, b2, eps,
                    b["I32"](*[s_["step"] for s_ in b["states"]]), scratch.data_ptr(), None, None, None, 0), "adam_multi_step")
        return loss
