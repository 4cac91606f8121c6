"""Test-view pose tracking and the FPS benchmark — the second caller of `render()` in the reference
(reference render.py:99-186, `render_set_optimize`): Gaussians frozen, one 7-vector pose per test view
optimised by Adam (lr 3e-3 for t, 1e-3 for q, weight_decay 1e-4, cosine annealing to 1e-4), masked L1 loss
(reference utils/loss_utils.py:17-23, mask = render > 0), best-loss pose kept.

Only `dL/dmeans3D` and `dL/drotations` leave the rasterizer here; the fused pose kernel reduces them to the
seven pose gradients on the device, so one tracking iteration is ~15 kernel launches instead of ~150.
"""
from __future__ import annotations

import time
from typing import List

import torch

from .gaussian_renderer import render
from .pose_utils import get_tensor_from_camera


def l1_loss_mask(network_output, gt, mask):
    return (torch.abs(network_output - gt) * mask).sum() / mask.sum()


def freeze_gaussians(gaussians):
    for t in (gaussians._xyz, gaussians._features_dc, gaussians._features_rest, gaussians._opacity, gaussians._scaling,
              gaussians._rotation):
        t.requires_grad_(False)


def optimize_view_pose(view, gaussians, pipe, background, init_pose: torch.Tensor | None = None, num_iter: int = 500):
    """Returns dict(pose=[7], initial_loss, best_loss, render=[3,H,W]) for one view."""
    dev = gaussians.get_xyz.device
    if init_pose is None:
        init_pose = get_tensor_from_camera(view.world_view_transform.transpose(0, 1).cpu())
    camera_pose = init_pose.detach().to(dev).float()
    cam_T = camera_pose[-3:].clone().requires_grad_()
    cam_q = camera_pose[:4].clone().requires_grad_()
    optimizer = torch.optim.Adam([{"params": [cam_T], "lr": 0.003}, {"params": [cam_q], "lr": 0.001}], betas=(0.9, 0.999),
                                 weight_decay=1e-4)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=num_iter, eta_min=0.0001)
    cand_q, cand_T = cam_q.clone().detach(), cam_T.clone().detach()
    best = torch.full((), 1e20, device=dev)
    gt = view.original_image[0:3].to(dev)
    initial_loss = None
    for it in range(num_iter):
        rendering = render(view, gaussians, pipe, background, camera_pose=torch.cat([cam_q, cam_T]))["render"]
        mask = (rendering > 0.0).float()
        loss = l1_loss_mask(rendering, gt, mask)
        loss.backward()
        with torch.no_grad():
            optimizer.step()
            optimizer.zero_grad(set_to_none=True)
            if it == 0:
                initial_loss = float(loss)
            # keep the best pose without a host round trip (the reference compares on the host every iteration)
            better = loss < best
            best = torch.where(better, loss.detach(), best)
            cand_q = torch.where(better, cam_q.detach(), cand_q)
            cand_T = torch.where(better, cam_T.detach(), cand_T)
        scheduler.step()
    pose = torch.cat([cand_q, cand_T])
    with torch.no_grad():
        final = render(view, gaussians, pipe, background, camera_pose=pose)["render"]
    return dict(pose=pose, initial_loss=initial_loss, best_loss=float(best), render=final)


def render_set_optimize(views: List, gaussians, pipe, background, num_iter: int = 500, init_poses=None):
    freeze_gaussians(gaussians)
    out = []
    for i, view in enumerate(views):
        out.append(optimize_view_pose(view, gaussians, pipe, background, None if init_poses is None else init_poses[i], num_iter))
    return out


def measure_fps(view, gaussians, pipe, background, pose, frames: int = 1000) -> dict:
    """reference render.py:172-186: `frames` renders of one view, sorted, the middle 80 % averaged.  The reference
    relies on its operator's internal blocking read-back for the timing to mean anything; here every frame is
    followed by an explicit synchronize."""
    dev = gaussians.get_xyz.device
    times = []
    with torch.no_grad():
        for _ in range(frames):
            t0 = time.perf_counter()
            render(view, gaussians, pipe, background, camera_pose=pose)
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)
            times.append(time.perf_counter() - t0)
    times.sort()
    lo, hi = frames // 10, frames - frames // 10
    mid = times[lo:hi] if hi > lo else times
    mean = sum(mid) / len(mid)
    return dict(fps=1.0 / mean, ms_per_frame=1e3 * mean)
