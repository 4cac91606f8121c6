"""Seeded synthetic inputs for tests and benchmarks.

There is no network, MASt3R checkpoint or dataset in the build/measure environment, so the
workloads BASELINE.json names are realised by two deterministic generators:

  syn_blob(P, W, H, seed)            random Gaussians in one camera's frustum (configs C2 / C4-blob)
  syn_pointmap(V, Wm, Hm, W, H, seed) V cameras on an arc looking at a smooth depth surface; one
                                     Gaussian per pointmap pixel, exactly as MASt3R's per-view
                                     pointmaps would seed them (configs C1', C3, C4, C5)

All randomness is drawn on the CPU from torch.Generator(seed) so CPU oracle and GPU see identical bits.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List

import torch

from .camera import Camera

SH_C0 = 0.28209479177387814


@dataclass
class BlobScene:
    means3D: torch.Tensor      # [P,3] camera frame
    scaling_logit: torch.Tensor  # [P,3] log-scales (pre-activation, like GaussianModel._scaling)
    rotation: torch.Tensor     # [P,4] (w,x,y,z), deliberately NOT unit length
    opacity_logit: torch.Tensor  # [P,1]
    features_dc: torch.Tensor  # [P,1,3]
    features_rest: torch.Tensor  # [P,15,3]
    camera: Camera
    bg: torch.Tensor

    @property
    def shs(self):
        return torch.cat([self.features_dc, self.features_rest], dim=1)


def syn_blob(P: int, W: int, H: int, seed: int = 0, opacity: str = "random", fovx_deg: float = 60.0,
             scale_mean: float = 0.02) -> BlobScene:
    g = torch.Generator().manual_seed(seed)
    fovx = math.radians(fovx_deg)
    tanx = math.tan(fovx / 2)
    tany = tanx * H / W
    fovy = 2 * math.atan(tany)
    u = lambda *s: torch.rand(*s, generator=g)
    n = lambda *s: torch.randn(*s, generator=g)
    z = 2.0 + 6.0 * u(P)
    behind = u(P) < 0.02  # exercise the z <= 0.2 cull
    z = torch.where(behind, -1.0 + 1.2 * u(P), z)
    x = (2.2 * u(P) - 1.1) * tanx * z
    y = (2.2 * u(P) - 1.1) * tany * z
    means = torch.stack([x, y, z], dim=1)
    scaling = math.log(scale_mean) + 0.6 * n(P, 3)
    q = n(P, 4)
    q = q / q.norm(dim=1, keepdim=True) * (0.9 + 0.2 * u(P, 1))
    if opacity == "random":
        op = 1.5 * n(P, 1)
    else:  # "init": what create_from_pcd gives every Gaussian (reference scene/gaussian_model.py:164)
        op = torch.full((P, 1), math.log(0.1 / 0.9))
    f_dc = (0.5 * n(P, 1, 3)) / SH_C0
    f_rest = 0.05 * n(P, 15, 3)
    cam = Camera(0, torch.eye(4), fovx, fovy, W, H)
    return BlobScene(means.float(), scaling.float(), q.float(), op.float(), f_dc.float(), f_rest.float(), cam,
                     torch.zeros(3))


# ------------------------------------------------------------------------------------------------
@dataclass
class PointmapScene:
    cameras: List[Camera]            # V train cameras (ground-truth poses)
    points: torch.Tensor             # [P,3] world
    colors: torch.Tensor             # [P,3] in [0,1]
    confidence: torch.Tensor         # [P,1] raw MASt3R-like confidence
    pose_noise_q: torch.Tensor       # [V,4] multiplicative quaternion perturbation for the student
    pose_noise_t: torch.Tensor       # [V,3]
    extent: float


def _look_at_w2c(eye: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    f = target - eye
    f = f / f.norm()
    up = torch.tensor([0.0, -1.0, 0.0])  # image y points down
    r = torch.linalg.cross(f, up)
    r = r / r.norm()
    d = torch.linalg.cross(f, r)
    Rwc = torch.stack([r, d, f], dim=0)  # rows = camera axes in world
    w2c = torch.eye(4)
    w2c[:3, :3] = Rwc
    w2c[:3, 3] = -Rwc @ eye
    return w2c


def syn_pointmap(V: int, Wm: int, Hm: int, W: int, H: int, seed: int = 0) -> PointmapScene:
    g = torch.Generator().manual_seed(seed)
    fovx = math.radians(60.0)
    tanx = math.tan(fovx / 2)
    tany = tanx * H / W
    fovy = 2 * math.atan(tany)
    yaws = torch.linspace(-10.0, 10.0, V) if V > 1 else torch.zeros(1)
    ph = 2 * math.pi * torch.rand(4, 3, generator=g)
    fr = 0.6 + 1.4 * torch.rand(4, 2, generator=g)
    amp = 0.25 * torch.rand(4, generator=g)
    cams, pts, cols = [], [], []
    fxm, fym = Wm / (2 * tanx), Hm / (2 * tany)
    vs, us = torch.meshgrid(torch.arange(Hm, dtype=torch.float32), torch.arange(Wm, dtype=torch.float32), indexing="ij")
    for v in range(V):
        a = math.radians(float(yaws[v]))
        eye = torch.tensor([5.0 * math.sin(a), 0.0, -5.0 * math.cos(a)])
        w2c = _look_at_w2c(eye, torch.zeros(3))
        cams.append(Camera(v, w2c, fovx, fovy, W, H))
        xn = (us + 0.5 - Wm / 2) / fxm
        yn = (vs + 0.5 - Hm / 2) / fym
        zc = 4.0 + sum(amp[k] * torch.sin(fr[k, 0] * 3 * xn + fr[k, 1] * 3 * yn + ph[k, v % 3]) for k in range(4))
        pc = torch.stack([xn * zc, yn * zc, zc], dim=-1).reshape(-1, 3)
        c2w = torch.linalg.inv(w2c)
        pw = pc @ c2w[:3, :3].t() + c2w[:3, 3]
        pts.append(pw)
        col = torch.stack([0.5 + 0.5 * torch.sin(3.0 * pw[:, 0] + 0.3), 0.5 + 0.5 * torch.sin(2.0 * pw[:, 1] + 1.1),
                           0.5 + 0.5 * torch.sin(2.5 * (pw[:, 0] + pw[:, 1]) + 2.0)], dim=-1)
        cols.append(col)
    points = torch.cat(pts).float()
    colors = torch.cat(cols).float().clamp(0, 1)
    conf = 3.0 + 2.0 * torch.randn(points.shape[0], 1, generator=g)
    # student pose perturbation: ~1 degree rotation, 0.02 translation
    ax = torch.randn(V, 3, generator=g)
    ax = ax / ax.norm(dim=1, keepdim=True)
    half = math.radians(1.0) / 2
    dq = torch.cat([torch.full((V, 1), math.cos(half)), ax * math.sin(half)], dim=1)
    dt = 0.02 * torch.randn(V, 3, generator=g)
    extent = float((points - points.mean(0)).norm(dim=1).max() * 1.1)
    return PointmapScene(cams, points, colors, conf.float(), dq.float(), dt.float(), extent)
