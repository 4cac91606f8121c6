"""ORACLE / CPU BASELINE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

One full train iteration (reference train.py:140-211) assembled from the CPU oracles: PyTorch glue
(pose transform, activations) + oracle/gs_ref.c rasterizer (forward + closed-form backward, OpenMP) +
the reference's PyTorch SSIM/L1 + the PerPointAdam restatement.  This is the "reference CPU path"
BASELINE.json asks to time next to the GPU (the reference itself has no CPU rasterizer);
bench.py reports it as cpu_baseline.kind = "port".
"""
import math
import random

import torch

from . import gs_ref
from .adam_ref import PerPointAdamRef
from .raster_torch import RasterSettings
from .ssim_ref import l1_loss, ssim


def _pose_to_w2c(pose):
    q = pose[:4] / pose[:4].norm()
    r, x, y, z = q
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]).reshape(3, 3)
    return R, pose[4:]


def _quadmul(q1, q2):
    w1, x1, y1, z1 = q1.unbind(-1)
    w2, x2, y2, z2 = q2.unbind(-1)
    return torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], dim=-1)


class CpuTrainer:
    """Mirrors instantsplat_amd.train on CPU tensors cloned from a TrainState-like parameter dict."""

    def __init__(self, params: dict, cameras, gt_images, per_point_lr, lrs: dict, lambda_dssim=0.2, sh_degree=0):
        self.p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in params.items()}
        self.cameras, self.gts = cameras, [g.detach().cpu() for g in gt_images]
        self.lam, self.deg = lambda_dssim, sh_degree
        groups = []
        for name in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "pose"):
            grp = {"params": [self.p[name]], "lr": lrs[name], "name": name}
            if name == "xyz" and per_point_lr is not None:
                grp["per_point_lr"] = per_point_lr.detach().cpu()
            groups.append(grp)
        self.opt = PerPointAdamRef(groups, lr=0, betas=(0.9, 0.999), eps=1e-15)
        self.rng = random.Random(0)
        self.stack = []

    def render(self, cam, pose):
        p = self.p
        R, t = _pose_to_w2c(pose)
        means = p["xyz"] @ R.t() + t
        rots = _quadmul(pose[:4], p["rotation"])
        st = RasterSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5),
                            torch.zeros(3), 1.0, torch.eye(4), cam.projection_matrix.cpu(), self.deg, torch.zeros(3), False, False)
        shs = torch.cat([p["f_dc"], p["f_rest"]], dim=1)
        m2d = torch.zeros_like(means, requires_grad=True)
        color, radii = gs_ref.rasterize(means, m2d, torch.sigmoid(p["opacity"]), st, shs=shs, scales=torch.exp(p["scaling"]),
                                        rotations=rots)
        return color

    def iteration(self):
        if not self.stack:
            self.stack = list(self.cameras)
        cam = self.stack.pop(self.rng.randint(0, len(self.stack) - 1))
        img = self.render(cam, self.p["pose"][cam.uid])
        gt = self.gts[cam.uid]
        loss = (1.0 - self.lam) * l1_loss(img, gt) + self.lam * (1.0 - ssim(img.unsqueeze(0), gt.unsqueeze(0)))
        loss.backward()
        val = loss.item()
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        return val
