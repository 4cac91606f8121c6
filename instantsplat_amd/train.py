"""Joint Gaussian + camera-pose optimisation loop — the body of reference train.py:124-227 on the
HIP path, driven by a synthetic 3-view pointmap scene (no MASt3R / dataset offline).

Per iteration (same order as the reference): LR schedule -> (SH degree up every 1000) -> pop a random
view -> render(camera_pose=P[uid]) -> (1-l)*L1 + l*(1-SSIM) -> backward -> loss.item() ->
optimizer.step() unless it is the last iteration -> zero_grad(set_to_none=True).
"""
from __future__ import annotations

import random
import time
from dataclasses import dataclass, field
from typing import List, Optional

import torch

from .arguments import ModelParams, OptimizationParams, PipelineParams
from .fused_ssim import fused_l1_ssim_loss, fused_ssim
from .gaussian_renderer import render
from .pose_utils import get_tensor_from_camera, quadmultiply
from .scene import GaussianModel, confidence_to_lr_modifiers
from .synthetic import PointmapScene


def l1_loss(a, b):
    return torch.abs(a - b).mean()


def psnr(img1, img2):
    """reference utils/image_utils.py:17-19 (per-row-of-first-dim MSE)."""
    mse = ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


@dataclass
class TrainState:
    gaussians: GaussianModel
    cameras: list
    gt_images: List[torch.Tensor]
    background: torch.Tensor
    opt: OptimizationParams
    pipe: PipelineParams
    iteration: int = 0
    viewpoint_stack: list = field(default_factory=list)
    rng: random.Random = field(default_factory=lambda: random.Random(0))
    last_loss: Optional[torch.Tensor] = None


def setup_training(scene: PointmapScene, device, opt: OptimizationParams | None = None, pipe: PipelineParams | None = None,
                   model: ModelParams | None = None) -> TrainState:
    """Teacher = create_from_pcd(scene points) at the true poses -> ground-truth images.
    Student = teacher with perturbed positions / colours / poses (what MASt3R noise would look like)."""
    opt = opt or OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True)
    pipe = pipe or PipelineParams()
    model = model or ModelParams()
    dev = torch.device(device)
    bg = torch.tensor([1.0, 1.0, 1.0] if model.white_background else [0.0, 0.0, 0.0], dtype=torch.float32, device=dev)
    g = torch.Generator().manual_seed(1234)

    teacher = GaussianModel(model.sh_degree)
    teacher.create_from_pcd(scene.points, scene.colors, scene.extent, dev)
    teacher.init_RT_seq(scene.cameras, dev)
    gts = []
    with torch.no_grad():
        for cam in scene.cameras:
            gts.append(render(cam, teacher, pipe, bg, camera_pose=teacher.get_RT(cam.uid))["render"].clamp(0, 1).detach())

    student = GaussianModel(model.sh_degree)
    noisy_pts = scene.points + 0.01 * torch.randn(scene.points.shape, generator=g)
    noisy_col = (scene.colors + 0.05 * torch.randn(scene.colors.shape, generator=g)).clamp(0, 1)
    student.create_from_pcd(noisy_pts, noisy_col, scene.extent, dev)
    student.init_RT_seq(scene.cameras, dev)
    with torch.no_grad():
        P = student.P.detach().clone()
        dq = scene.pose_noise_q.to(dev)
        P[:, :4] = quadmultiply(dq, P[:, :4])
        P[:, 4:] += scene.pose_noise_t.to(dev)
    student.P = P.requires_grad_(True)
    conf_lr = confidence_to_lr_modifiers(scene.confidence.to(dev), scale=(1.0, 100.0))
    if opt.pp_optimizer:
        student.training_setup_pp(opt, conf_lr)
    else:
        student.training_setup(opt)
    return TrainState(student, list(scene.cameras), gts, bg, opt, pipe)


def train_iteration(st: TrainState, fused_loss: bool = True, sync_loss: bool = True):
    """One pass of reference train.py:140-211. Returns the loss (python float if sync_loss)."""
    st.iteration += 1
    it, g, opt = st.iteration, st.gaussians, st.opt
    g.update_learning_rate(it)
    if not opt.optim_pose:
        g.P.requires_grad_(False)
    if it % 1000 == 0:
        g.oneupSHdegree()
    if not st.viewpoint_stack:
        st.viewpoint_stack = list(st.cameras)
    cam = st.viewpoint_stack.pop(st.rng.randint(0, len(st.viewpoint_stack) - 1))
    pose = g.get_RT(cam.uid)
    bg = torch.rand(3, device=st.background.device) if opt.random_background else st.background
    pkg = render(cam, g, st.pipe, bg, camera_pose=pose)
    image = pkg["render"]
    gt = st.gt_images[cam.uid]
    if fused_loss:
        loss, _ = fused_l1_ssim_loss(image.unsqueeze(0), gt.unsqueeze(0), opt.lambda_dssim)
    else:
        Ll1 = l1_loss(image, gt)
        loss = (1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - fused_ssim(image.unsqueeze(0), gt.unsqueeze(0)))
    loss.backward()
    out = loss.item() if sync_loss else loss.detach()
    with torch.no_grad():
        if it < opt.iterations:
            g.optimizer.step()
            g.optimizer.zero_grad(set_to_none=True)
    st.last_loss = out
    return out


@torch.no_grad()
def evaluate_psnr(st: TrainState) -> float:
    vals = []
    for cam in st.cameras:
        img = render(cam, st.gaussians, st.pipe, st.background, camera_pose=st.gaussians.get_RT(cam.uid))["render"].clamp(0, 1)
        vals.append(psnr(img, st.gt_images[cam.uid]).mean())
    return float(torch.stack(vals).mean())


def training(scene: PointmapScene, device, iterations: int = 1000, log_every: int = 0, **kw) -> dict:
    opt = OptimizationParams(iterations=iterations, pp_optimizer=True, optim_pose=True)
    st = setup_training(scene, device, opt=opt)
    psnr0 = evaluate_psnr(st)
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = []
    for i in range(iterations):
        losses.append(train_iteration(st, **kw))
        if log_every and (i + 1) % log_every == 0:
            print(f"[iter {i + 1}] loss {losses[-1]:.6f}")
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(seconds=dt, iters_per_sec=iterations / dt, first_loss=losses[0], last_loss=losses[-1], psnr_before=psnr0,
                psnr_after=evaluate_psnr(st), state=st)
