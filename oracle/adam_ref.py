"""ORACLE — TEST INFRASTRUCTURE.  Pure-PyTorch restatement of the update rule of the reference's
PerPointAdam (reference scene/per_point_adam.py:34-100), pinned by the trajectory in
tests/golden/reference_vectors.npz that was produced by the reference class itself.

  mask  = ||grad|| > 0                      (ONE boolean for the whole tensor)
  if mask: m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
  denom = sqrt(v) + eps
  step  = lr * sqrt(1-b2^t) / (1-b1^t)
  p    -= step * per_point_lr * m / denom   (per_point_lr broadcast over the row; constant)
"""
import torch


class PerPointAdamRef(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, per_point_lr=None))

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            pplr = group.get("per_point_lr")
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                st = self.state[p]
                if not st:
                    st["step"], st["m"], st["v"] = 0, torch.zeros_like(p), torch.zeros_like(p)
                st["step"] += 1
                if group["weight_decay"] != 0:
                    g = g + group["weight_decay"] * p
                if bool(g.norm() > 0):
                    st["m"].mul_(b1).add_(g, alpha=1 - b1)
                    st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = st["v"].sqrt().add_(group["eps"])
                step = group["lr"] * ((1 - b2 ** st["step"]) ** 0.5 / (1 - b1 ** st["step"]))
                upd = st["m"] / denom
                p.add_(-(step * pplr) * upd if pplr is not None else -step * upd)
