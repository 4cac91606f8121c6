"""BASELINE.json configs[0] ("3-view sora/50 iters plumbing") on the reference's own example frames — test infrastructure.

The reference gets there with `init_geo.py` (MASt3R: a checkpoint that cannot be obtained offline) followed by `train.py -s
assets/sora/Art -n 3`.  Here the init stage is REPLACED — plainly: nothing below comes from MASt3R — by
  * the three frames of reference assets/sora/Art/images, re-encoded at their native 1280 x 720 (tests/golden/sora_art/*.jpg,
    tests/golden/make_golden_sora.py),
  * a synthetic pointmap per view (instantsplat_amd.synthetic.syn_pointmap's smooth depth on a Wm x Hm grid, cameras on an arc,
    field of view 60 degrees) whose points take the colour of the frame's pixel they project to, as MASt3R's do,
  * noisy poses / a random confidence map, as syn_pointmap's student gets them,
written to disk in the layout the init stage leaves (scene_io.write_init_scene, the JPEG files copied as they are).  From there
everything is the product's path: scene_io.load_init_scene (JPEG decode, `-r` resizing, non-square 16:9 frames), train."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
FRAMES = [os.path.join(HERE, "golden", "sora_art", n) for n in ("art_frame_0.jpg", "art_frame_1.jpg", "art_frame_2.jpg")]


def write_sora_init_dir(dst: str, Wm: int = 160, Hm: int = 90, seed: int = 0):
    """-> (W, H) of the frames.  3 x Wm x Hm points."""
    from PIL import Image
    from instantsplat_amd import scene_io
    from instantsplat_amd.pose_utils import get_camera_from_tensor, get_tensor_from_camera, quadmultiply
    from instantsplat_amd.synthetic import syn_pointmap
    with Image.open(FRAMES[0]) as im:
        W, H = im.size
    sc = syn_pointmap(len(FRAMES), Wm, Hm, W, H, seed=seed)
    cols = []
    for f in FRAMES:   # the colour of a point = the frame at the pixel it was "measured" at (box-filtered to the pointmap grid)
        with Image.open(f) as im:
            small = np.asarray(im.convert("RGB").resize((Wm, Hm), Image.BOX), dtype=np.float32) / 255.0
        cols.append(torch.from_numpy(small).reshape(-1, 3))
    w2c = []
    for v, c in enumerate(sc.cameras):   # the estimated (noisy) poses an initialisation would hand over
        p = get_tensor_from_camera(c.world_view_transform.t())
        p = torch.cat([quadmultiply(sc.pose_noise_q[v:v + 1], p[None, :4])[0], p[4:] + sc.pose_noise_t[v]])
        w2c.append(get_camera_from_tensor(p).double().numpy())
    g = torch.Generator().manual_seed(seed + 7)
    pts = sc.points + 0.01 * torch.randn(sc.points.shape, generator=g)
    scene_io.write_init_scene(dst, w2c, [(c.FoVx, c.FoVy) for c in sc.cameras], FRAMES, pts, torch.cat(cols), sc.confidence,
                              names=["%d.jpg" % v for v in range(len(FRAMES))])   # the reference's own file names (assets/sora/Art/images/0.jpg ...)
    return W, H


def train_against_cpu_oracle(scene, dev, iters, fused_loss=False):
    """`iters` iterations of the drop-in loop (train.py's loss as written) on `dev` and of the all-CPU oracle trainer from the same
    state, same view order -> (device losses, oracle losses, state)"""
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.train import setup_training, train_iteration
    from oracle.train_ref import CpuTrainer
    st = setup_training(scene, dev, opt=OptimizationParams(iterations=10 ** 6, pp_optimizer=True, optim_pose=True))
    g = st.gaussians
    g.update_learning_rate(1)
    lrs = {grp["name"]: grp["lr"] for grp in g.optimizer.param_groups}
    params = dict(xyz=g._xyz, f_dc=g._features_dc, f_rest=g._features_rest, opacity=g._opacity, scaling=g._scaling, rotation=g._rotation, pose=g.P)
    cpu = CpuTrainer(params, st.cameras, st.gt_images, g.per_point_lr, lrs)
    cpu.rng.setstate(st.rng.getstate())   # the view sampling continues on the stream the camera shuffle drew from
    l_dev, l_cpu = [], []
    for _ in range(iters):
        l_dev.append(train_iteration(st, fused_loss=fused_loss))
        for grp, dgrp in zip(cpu.opt.param_groups, g.optimizer.param_groups):
            grp["lr"] = dgrp["lr"]
        l_cpu.append(cpu.iteration())
    return l_dev, l_cpu, st
