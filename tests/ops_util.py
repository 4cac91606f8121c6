"""Checks shared by the CPU (emulated kernels) and GPU tiers for SSIM / kNN / Adam / training."""
import os

import numpy as np
import torch

from oracle import knn_ref, ssim_ref

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))
T = lambda k: torch.from_numpy(G[k])

# ---- measured bounds.  Every device-vs-device / device-vs-oracle tolerance below goes through `bound`, which records the
# measured value; `GS_CALIBRATE=1 pytest ... -s` prints them all at the end of the session instead of failing (conftest.py),
# which is how the GPU-tier limits were set (round 2: measured on MI355X, limit = roughly 10-30x the measurement, never a guess).
MEASURED = []


def bound(label, value, limit):
    value = float(value)
    MEASURED.append((label, value, float(limit)))
    if os.environ.get("GS_CALIBRATE") != "1":
        assert value <= limit, (label, value, limit)


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def generic_start(st, seed=101):
    """Moves a freshly set-up student off the two measure-zero sets the reference's `create_from_pcd` state sits on, which
    make trajectory comparisons meaningless for reasons that have nothing to do with the code under test:
      * isotropic scales + identity rotations: d(loss)/d(rotation) is exactly zero, only rounding noise is computed, and
        Adam (eps 1e-15) turns noise into +-lr steps;
      * colour channels clamped to exactly 0 or 1: f_dc = RGB2SH(0) puts C0 * f_dc + 0.5 on the edge of the SH colour
        clamp, where the gradient mask depends on the last bit (round 2, tools/diag_step0.py).
    Same treatment as the recorded reference run gets in tests/golden/make_golden.py (_Scene)."""
    g = st.gaussians
    gen = torch.Generator().manual_seed(seed)
    n = g._xyz.shape[0]
    dev = g._xyz.device
    with torch.no_grad():
        g._scaling.add_((0.35 * torch.randn(n, 3, generator=gen)).to(dev))
        q = torch.randn(n, 4, generator=gen)
        q = q / q.norm(dim=1, keepdim=True) * (0.9 + 0.2 * torch.rand(n, 1, generator=gen))
        g._rotation.copy_(q.to(dev))
        g._features_dc.add_((0.02 + 0.02 * torch.rand(g._features_dc.shape, generator=gen)).to(dev)
                            * torch.where(g._features_dc < 0, 1.0, -1.0))
    return st


def check_ssim_golden(dev):
    from instantsplat_amd.fused_ssim import fused_l1_ssim_loss, fused_ssim
    for name in ("a", "b"):
        x = T(f"ssim_{name}_x").to(dev).requires_grad_(True)
        y = T(f"ssim_{name}_y").to(dev)
        v = fused_ssim(x, y)
        v.backward()
        assert abs(float(v) - float(G[f"ssim_{name}_val"])) <= 1e-6, (float(v), float(G[f"ssim_{name}_val"]))
        g_ref = T(f"ssim_{name}_grad")
        rel = float((x.grad.cpu() - g_ref).norm() / g_ref.norm())
        assert rel <= 1e-5, rel  # stated tolerance: SSIM |d| <= 1e-6, gradient rel-L2 <= 1e-5
        x.grad = None
        loss, parts = fused_l1_ssim_loss(x, y, 0.2)
        loss.backward()
        ref_loss = 0.8 * float(G[f"l1_{name}_val"]) + 0.2 * (1.0 - float(G[f"ssim_{name}_val"]))
        assert abs(float(loss) - ref_loss) <= 1e-6
        g2 = 0.8 * T(f"l1_{name}_grad") - 0.2 * g_ref
        assert float((x.grad.cpu() - g2).norm() / g2.norm()) <= 1e-5


def check_ssim_random(dev, H, W, seed=0, padding="same"):
    from instantsplat_amd.fused_ssim import fused_ssim
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(1, 3, H, W, generator=g)
    y = (x + 0.2 * torch.randn(1, 3, H, W, generator=g)).clamp(0, 1)
    xr = x.clone().requires_grad_(True)
    vr = ssim_ref.ssim(xr, y, padding=padding)
    vr.backward()
    xd = x.to(dev).requires_grad_(True)
    vd = fused_ssim(xd, y.to(dev), padding=padding)
    vd.backward()
    assert abs(float(vd) - float(vr)) <= 1e-6
    assert float((xd.grad.cpu() - xr.grad).norm() / xr.grad.norm()) <= 1e-5


def check_knn(dev, n, seed=0, duplicates=False):
    from instantsplat_amd.simple_knn._C import distCUDA2
    g = torch.Generator().manual_seed(seed)
    pts = torch.randn(n, 3, generator=g) * torch.tensor([3.0, 1.0, 0.2])
    if duplicates and n > 10:
        pts[5] = pts[4]
        pts[9] = pts[4]
    ref = knn_ref.dist2(pts)
    out = distCUDA2(pts.to(dev)).cpu()
    assert out.shape == (n,)
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-9), float((out - ref).abs().max())  # stated tolerance: rel 1e-5


def check_knn_pointmap(dev, V, Wm):
    """distCUDA2 on the point clouds the BASELINE configs start from (C3: 3 x 256^2 = 196,608 points; C4: 12 x 288^2 = 995,328):
    back-projected depth maps — locally grid-like, overlapping views, strongly anisotropic density — against the float64 k-d tree."""
    from instantsplat_amd.simple_knn._C import distCUDA2
    from instantsplat_amd.synthetic import syn_pointmap
    pts = syn_pointmap(V, Wm, Wm, 64, 64, seed=0).points.float()
    assert pts.shape == (V * Wm * Wm, 3)
    ref = knn_ref.dist2(pts)
    out = distCUDA2(pts.to(dev)).cpu()
    assert out.shape == ref.shape
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-9), float(((out - ref).abs() / ref.clamp_min(1e-12)).max())  # rel 1e-5


def check_adam_golden(dev):
    from instantsplat_amd.optim import PerPointAdam
    p1, p2 = T("adam_p1_0").clone().to(dev).requires_grad_(True), T("adam_p2_0").clone().to(dev).requires_grad_(True)
    opt = PerPointAdam([{"params": [p1], "per_point_lr": T("adam_pplr").to(dev), "lr": 1.6e-4, "name": "xyz"},
                        {"params": [p2], "lr": 2.5e-2, "name": "f_dc"}], lr=0, betas=(0.9, 0.999), eps=1e-15, weight_decay=0.0)
    for t in range(4):
        p1.grad, p2.grad = T("adam_g1")[t].clone().to(dev), T("adam_g2")[t].clone().to(dev)
        opt.step()
        assert torch.allclose(p1.detach().cpu(), T(f"adam_p1_{t + 1}"), rtol=2e-6, atol=1e-7), t
        assert torch.allclose(p2.detach().cpu(), T(f"adam_p2_{t + 1}"), rtol=2e-6, atol=1e-7), t


def oracle_frame_grads(params, cam, gt, dtype):
    """Gradients of the training loss of ONE frame from the CPU oracle in `dtype` (float32: what oracle/train_ref.CpuTrainer
    computes; float64: the yardstick), from the raw parameters `params` (name -> tensor) through the reference's glue —
    pose -> camera-frame means and rotations, activations, rasterizer, 0.8 L1 + 0.2 (1 - SSIM)."""
    import math
    from oracle import gs_ref
    from oracle.raster_torch import RasterSettings
    from oracle.ssim_ref import l1_loss, ssim
    from oracle.train_ref import _pose_to_w2c, _quadmul
    p = {k: v.detach().cpu().to(dtype).clone().requires_grad_(True) for k, v in params.items()}
    pose = p["pose"][cam.uid]
    R, t = _pose_to_w2c(pose)
    means = p["xyz"] @ R.t() + t
    rots = _quadmul(pose[:4], p["rotation"])
    st = RasterSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3, dtype=dtype), 1.0,
                        torch.eye(4, dtype=dtype), cam.projection_matrix.cpu().to(dtype), 0, torch.zeros(3, dtype=dtype), False, False)
    img, _ = gs_ref.rasterize(means, torch.zeros_like(means, requires_grad=True), torch.sigmoid(p["opacity"]), st,
                              shs=torch.cat([p["f_dc"], p["f_rest"]], dim=1), scales=torch.exp(p["scaling"]), rotations=rots)
    g = gt.detach().cpu().to(dtype)
    (0.8 * l1_loss(img, g) + 0.2 * (1.0 - ssim(img.unsqueeze(0), g.unsqueeze(0)))).backward()
    return {k: v.grad for k, v in p.items()}


def check_train_matches_cpu_oracle(dev, iters, Wm=16, W=32, fused_step=False):
    """Full train iterations on the device path vs the all-CPU oracle trainer from identical state
    (fused_step: the iterations of part (2) run on the one-call library step instead of the op-by-op path).

    (1) gradients of one iteration agree per tensor (rel-L2 <= 1e-4 against the fp32 oracle; a tensor outside that is judged
        against the float64 oracle: at most 2.5 x the fp32 oracle's own error).  `rotation` is compared with an
        absolute bound instead: at initialisation every Gaussian is isotropic (3 equal scales, reference
        scene/gaussian_model.py:160), so d(loss)/d(rotation) is mathematically zero and both sides hold
        only rounding noise — which Adam then turns into +-lr steps, in the reference as well.
    (2) the loss trajectories of `iters` full iterations (render, loss, backward, PerPointAdam) agree."""
    from instantsplat_amd.fused_ssim import fused_l1_ssim_loss
    from instantsplat_amd.gaussian_renderer import render
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import setup_training, train_iteration
    from oracle.ssim_ref import l1_loss, ssim
    from oracle.train_ref import CpuTrainer
    sc = syn_pointmap(3, Wm, Wm, W, W, seed=3)
    st = setup_training(sc, dev)
    g = st.gaussians
    params = dict(xyz=g._xyz, f_dc=g._features_dc, f_rest=g._features_rest, opacity=g._opacity, scaling=g._scaling,
                  rotation=g._rotation, pose=g.P)
    g.update_learning_rate(1)
    lrs = {grp["name"]: grp["lr"] for grp in g.optimizer.param_groups}
    cpu = CpuTrainer(params, st.cameras, st.gt_images, g.per_point_lr, lrs)

    cam = st.cameras[1]
    img = render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(cam.uid))["render"]
    loss, _ = fused_l1_ssim_loss(img.unsqueeze(0), st.gt_images[cam.uid].unsqueeze(0), 0.2)
    loss.backward()
    img_c = cpu.render(cam, cpu.p["pose"][cam.uid])
    gt_c = cpu.gts[cam.uid]
    loss_c = 0.8 * l1_loss(img_c, gt_c) + 0.2 * (1.0 - ssim(img_c.unsqueeze(0), gt_c.unsqueeze(0)))
    loss_c.backward()
    assert abs(float(loss) - float(loss_c)) <= 1e-6
    assert float((img.detach().cpu() - img_c.detach()).abs().max()) <= 1e-4
    gscale = max(float(cpu.p[k].grad.abs().max()) for k in ("xyz", "scaling", "opacity"))
    g64 = None
    for name, t in params.items():
        a, b = t.grad.detach().cpu(), cpu.p[name].grad
        if name == "rotation":
            assert float((a - b).abs().max()) <= 1e-5 * gscale, name
        elif float(b.norm()) > 0 and float((a - b).norm() / b.norm()) > 1e-4:
            # Outside the small-size criterion against the fp32 oracle: whose error is it?  The same frame through the SAME oracle
            # in float64 decides (BASELINE.md 3.1's oracle-relative contract): one pixel whose |image - gt| is below rounding flips
            # the sign of its L1 term in one fp32 implementation and not in the other, which moves every gradient of a 2000-Gaussian
            # scene by 1e-3 (tools/fuzz_ops.py seed 41, Wm 27 / W 96: the device stood 1.9e-5 from float64, the fp32 ORACLE 1.1e-3;
            # profiles/r06_diag_train_case_seed41.txt).
            if g64 is None:
                g64 = oracle_frame_grads(params, cam, st.gt_images[cam.uid], torch.float64)
            e_dev = float((a.double() - g64[name]).norm() / g64[name].norm())
            e_ref = float((b.double() - g64[name]).norm() / g64[name].norm())
            assert e_dev <= max(1e-4, 2.5 * e_ref), (name, "device vs fp64", e_dev, "fp32 oracle vs fp64", e_ref)
        t.grad = None
        cpu.p[name].grad = None

    if fused_step:
        from instantsplat_amd.train import FusedTrainer
        assert FusedTrainer.supported(st)
    for it in range(iters):
        l_dev = train_iteration(st, fused_step=fused_step)
        assert (getattr(st, "_trainer", None) is not None) == fused_step
        for grp, dgrp in zip(cpu.opt.param_groups, g.optimizer.param_groups):
            grp["lr"] = dgrp["lr"]
        l_cpu = cpu.iteration()
        # this test starts from the reference's isotropic create_from_pcd state ON PURPOSE (part (1) checks the structurally
        # zero rotation gradient), so Adam turns the rotations' rounding noise into +-lr steps on both sides and the losses
        # separate slowly; the tight trajectory checks are the generic-start and reference-function tests
        bound("cpu_oracle_loop/loss[%s]" % ("one-call" if fused_step else "op-by-op"), abs(l_dev - l_cpu) / max(1e-2, abs(l_cpu)), 5e-3)


def check_pose_activations(dev, P=777, seed=0):
    """Fused pose transform + activations (fwd and bwd incl. the 7 pose gradients) vs op-by-op PyTorch autograd
    with the reference's semantics (normalised quaternion for the means, raw for the Hamilton product)."""
    from instantsplat_amd.fused import pose_activations
    from instantsplat_amd.pose_utils import get_camera_from_tensor, quadmultiply
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    base = dict(xyz=rn(P, 3), rot=rn(P, 4), scaling=0.3 * rn(P, 3) - 3.0, opl=rn(P, 1), pose=torch.cat([rn(4) * 1.3, rn(3)]))
    w = [rn(P, 3), rn(P, 4), rn(P, 3), rn(P, 1)]
    res = {}
    for which in ("ref", "dut"):
        d = torch.device("cpu") if which == "ref" else torch.device(dev)
        t = {k: v.clone().to(d).requires_grad_(True) for k, v in base.items()}
        if which == "ref":
            M = get_camera_from_tensor(t["pose"])
            outs = (t["xyz"] @ M[:3, :3].t() + M[:3, 3], quadmultiply(t["pose"][:4], t["rot"]), torch.exp(t["scaling"]),
                    torch.sigmoid(t["opl"]))
        else:
            outs = pose_activations(t["xyz"], t["rot"], t["scaling"], t["opl"], t["pose"])
        sum((o * wi.to(d)).sum() for o, wi in zip(outs, w)).backward()
        res[which] = ([o.detach().cpu() for o in outs], {k: v.grad.detach().cpu() for k, v in t.items()})
    for a, b in zip(res["dut"][0], res["ref"][0]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    for k in base:
        a, b = res["dut"][1][k], res["ref"][1][k]
        assert float((a - b).norm() / (b.norm() + 1e-30)) <= 1e-5, k


def check_fused_render_equals_unfused(dev, degree=0):
    """The three glues of render() — one posed autograd node (default), the round-1 fused glue (pose/activation kernel + SH
    view + operator) and the op-by-op PyTorch glue, all on the HIP rasterizer — give the same image and the same gradients
    for all seven tensors, at SH degree 0 (f_rest gets an all-zero gradient) and above (f_rest is differentiated)."""
    import instantsplat_amd.gaussian_renderer as gr
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import setup_training
    sc = syn_pointmap(3, 24, 24, 64, 64, seed=5)
    st = generic_start(setup_training(sc, dev))
    g = st.gaussians
    g.active_sh_degree = degree
    if degree:
        gen = torch.Generator().manual_seed(3)
        with torch.no_grad():
            g._features_rest.copy_(0.05 * torch.randn(g._features_rest.shape, generator=gen).to(dev))
    names = dict(xyz=g._xyz, f_dc=g._features_dc, f_rest=g._features_rest, opacity=g._opacity, scaling=g._scaling,
                 rotation=g._rotation, pose=g.P)
    out = {}
    default = gr.FUSED_GLUE
    assert default == "posed"
    for fused in ("posed", True, False):
        gr.FUSED_GLUE = fused
        try:
            for t in names.values():
                t.grad = None
            cam = st.cameras[2]
            pkg = gr.render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(cam.uid))
            img = pkg["render"]
            (img * st.gt_images[cam.uid]).sum().backward()
            out[fused] = (img.detach().cpu(), {k: (None if t.grad is None else t.grad.detach().cpu().clone()) for k, t in names.items()},
                          pkg["radii"].cpu(), pkg["viewspace_points"].grad.detach().cpu().clone())
        finally:
            gr.FUSED_GLUE = default
    cuda = torch.device(dev).type == "cuda"
    for fused in ("posed", True):
        assert torch.equal(out[fused][2], out[False][2]), fused
        bound("render_glues/image[%s]" % fused, (out[fused][0] - out[False][0]).abs().max(), 1e-5)
        bound("render_glues/grad_viewspace_points[%s]" % fused, rel_l2(out[fused][3], out[False][3]), 1e-4)
        for k in names:
            a, b = out[fused][1][k], out[False][1][k]
            if float(b.norm()) == 0:
                assert float(a.norm()) == 0, (fused, k)
            else:
                bound("render_glues/grad_%s[%s]" % (k, fused), rel_l2(a, b), 1e-4)
    assert (float(out["posed"][1]["f_rest"].abs().max()) > 0) == (degree > 0)
    if not cuda:   # the posed node and the three-node glue run the same kernels' arithmetic: bit for bit under the emulator
        assert torch.equal(out["posed"][0], out[True][0])


def check_run_ahead_equals_sync_loop(dev, iters=23, force_overflow=False, Wm=20, W=48):
    """The sync-free driver (device loss ring, bounded instance buffers, snapshot/replay on overflow) must
    reproduce the synchronous loop: same parameters and the same EMA.  `force_overflow` shrinks the capacity
    below the true instance count so every window is detected as overflowed and replayed in exact mode."""
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import RunAhead, setup_training, train_iteration
    sc = syn_pointmap(3, Wm, Wm, W, W, seed=7)
    mk = lambda: generic_start(setup_training(sc, dev, opt=OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True)))
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P")
    try:
        st_a = mk()
        ema = 0.0
        for _ in range(iters):
            ema = 0.4 * train_iteration(st_a) + 0.6 * ema
        st_b = mk()
        if force_overflow:
            BinningPolicy.slack, BinningPolicy.pad = 0.5, 0
        ra = RunAhead(st_b, window=5 if iters > 10 else 2)
        for _ in range(iters):
            ra.step()
        ema_b = ra.flush()
        if force_overflow:
            assert ra.replays >= 3, ra.replays
        # CPU tier: the emulated kernels are deterministic and the one-call step evaluates the same expressions as the
        # op-by-op path, so everything must agree to rounding.  GPU: the two paths round differently (the one-call step
        # transforms by the pose inside the projection kernels) and float-atomic order differs between any two runs; from the
        # generic start (no structurally zero gradients, no colours on the clamp edge) that stays at rounding level — the
        # 1e-2-level separations round 1 bounded here came from the degenerate start, not from the loops.
        cuda = torch.device(dev).type == "cuda"
        bound("run_ahead_vs_sync/ema", abs(ema - ema_b) / max(1e-3, abs(ema)), 2e-3 if cuda else 1e-6)   # MI355X: 1.1e-4 (23 iterations)
        for n in names:
            bound("run_ahead_vs_sync/param" + n, rel_l2(getattr(st_b.gaussians, n), getattr(st_a.gaussians, n)), 2e-3 if cuda else 1e-5)   # MI355X: <= 1.3e-4
    finally:
        BinningPolicy.slack, BinningPolicy.pad = 1.5, 16384
        BinningPolicy.reset("exact")


def check_run_ahead_ring_stays_a_leaf_with_the_loss_as_written(dev, Wm=10, W=24):
    """RunAhead on the autograd path with train.py's loss lines as written (fused_loss=False, no one-call
    trainer).  `_forward_backward_step` hands back the LazyScalar; written into the persistent loss ring as it is, its materialised
    tensor — which requires grad after the backward — turned the ring into a non-leaf and chained every iteration's render graph
    onto it for the whole run.  The ring must stay a plain buffer, and hold the same values as the per-iteration read."""
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import RunAhead, setup_training, train_iteration
    sc = syn_pointmap(3, Wm, Wm, W, W, seed=3)
    mk = lambda: setup_training(sc, dev, opt=OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True))
    try:
        st_a = mk()
        want = [train_iteration(st_a, fused_loss=False) for _ in range(4)]
        ra = RunAhead(mk(), window=8, fused_loss=False, fused_step=False)
        assert ra.trainer is None
        for _ in range(4):
            ra.step()
            assert ra.ring.grad_fn is None and not ra.ring.requires_grad
        got = ra.ring[:4].tolist()
        ra.flush()
        cuda = torch.device(dev).type == "cuda"
        for a, b in zip(got, want):
            bound("run_ahead_ring/loss", abs(a - b) / max(1e-3, abs(b)), 2e-3 if cuda else 1e-6)
    finally:
        BinningPolicy.reset("exact")


def check_run_ahead_sticky_commit_gate(dev, iters=17, Wm=20, W=48):
    """An overflow in the MIDDLE of a run-ahead window on the one-call step: the buffers hold every view but the heaviest one, so
    windows overflow at whatever position that view is drawn.  The sticky commit gate (include/mi355gs.h) makes that iteration's
    optimizer launch and every one enqueued behind it a no-op; RunAhead keeps the iterations before it, rewinds only its host
    half and redoes the rest — no snapshot of parameters or moments exists on this path.  Must equal the synchronous loop."""
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import RunAhead, hint_key, setup_training, train_iteration
    sc = syn_pointmap(3, Wm, Wm, W, W, seed=7)
    mk = lambda: generic_start(setup_training(sc, dev, opt=OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True)))
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P")
    try:
        st_a = mk()
        ema = 0.0
        for _ in range(iters):
            ema = 0.4 * train_iteration(st_a) + 0.6 * ema
        st_b = mk()
        ra = RunAhead(st_b, window=6)
        assert ra.trainer is not None and ra.snap is None        # one-call step, nothing to restore from
        counts = sorted(BinningPolicy.known[hint_key(st_b, c)] for c in st_b.cameras)
        assert counts[-1] > counts[-2] + 8, counts               # the scene must have one strictly heaviest view
        # every (re)made handle holds the second-heaviest view with a little room, not the heaviest
        BinningPolicy.slack, BinningPolicy.pad = (counts[-2] + (counts[-1] - counts[-2]) * 0.5) / counts[-1], 0
        ra._make_trainer()
        for _ in range(iters):
            ra.step()
        ema_b = ra.flush()
        assert ra.replays >= 1 and ra.partial_replays >= 1, (ra.replays, ra.partial_replays)
        cuda = torch.device(dev).type == "cuda"
        bound("run_ahead_sticky/ema", abs(ema - ema_b) / max(1e-3, abs(ema)), 2e-3 if cuda else 1e-6)
        for n in names:
            bound("run_ahead_sticky/param" + n, rel_l2(getattr(st_b.gaussians, n), getattr(st_a.gaussians, n)), 2e-3 if cuda else 1e-5)
        assert st_b.iteration == st_a.iteration
    finally:
        BinningPolicy.slack, BinningPolicy.pad = 1.5, 16384
        BinningPolicy.reset("exact")


def check_pose_tracking(dev, num_iter, Wm=16, W=40, min_gain=0.0):
    """render_set_optimize (reference render.py:99-170): Gaussians frozen, a perturbed view pose is pulled back
    towards the pose that explains the image (masked L1)."""
    from instantsplat_amd.pose_tracking import measure_fps, render_set_optimize
    from instantsplat_amd.pose_utils import get_tensor_from_camera
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import setup_training
    sc = syn_pointmap(3, Wm, Wm, W, W, seed=11)
    st = setup_training(sc, dev)
    g = st.gaussians
    view = st.cameras[1]
    view.original_image = st.gt_images[1]
    true_pose = get_tensor_from_camera(view.world_view_transform.transpose(0, 1).cpu())
    init = true_pose.clone()
    init[4:] += torch.tensor([0.03, -0.02, 0.04])
    res = render_set_optimize([view], g, st.pipe, st.background, num_iter=num_iter, init_poses=[init])[0]
    assert all(not t.requires_grad for t in (g._xyz, g._features_dc, g._opacity))
    assert res["best_loss"] <= res["initial_loss"] * (1.0 - min_gain), (res["initial_loss"], res["best_loss"])
    assert res["render"].shape == (3, W, W)
    fps = measure_fps(view, g, st.pipe, st.background, res["pose"], frames=5)
    assert fps["fps"] > 0
    return res


def check_fused_train_step_equals_autograd_path(dev, iters=6, Wm=12, W=32):
    """mi355gs_trainer_step (one call per iteration) vs the op-by-op autograd path: same kernels, same results."""
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import RunAhead, setup_training
    sc = syn_pointmap(3, Wm, Wm, W, W, seed=9)
    mk = lambda: generic_start(setup_training(sc, dev, opt=OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True)))
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P")
    try:
        res = {}
        for fused in (False, True):
            st = mk()
            ra = RunAhead(st, window=3, fused_step=fused)
            assert (ra.trainer is not None) == fused
            for _ in range(iters):
                ra.step()
            res[fused] = (ra.flush(), {n: getattr(st.gaussians, n).detach().cpu().clone() for n in names},
                          {n: st.gaussians.optimizer.state[getattr(st.gaussians, n)]["exp_avg_sq"].detach().cpu().clone() for n in names})
            BinningPolicy.reset("exact")
        cuda = torch.device(dev).type == "cuda"
        bound("one_call_vs_autograd/ema", abs(res[True][0] - res[False][0]) / max(1e-3, abs(res[False][0])), 2e-4 if cuda else 1e-6)
        # Under the emulator (deterministic atomics) the two paths agree to rounding; on the GPU to float-atomic order.  The
        # second moments (sums of squared gradients) are the most sensitive quantity compared here.
        for n in names:
            for k, what in ((1, "param"), (2, "exp_avg_sq")):
                # MI355X: parameters <= 1.0e-5, second moments <= 1.4e-4
                bound("one_call_vs_autograd/%s%s" % (what, n), rel_l2(res[True][k][n], res[False][k][n]),
                      ((2e-4 if k == 1 else 2e-3) if cuda else 1e-5))
    finally:
        BinningPolicy.reset("exact")


def check_fused_step_gradients_equal_autograd(dev, degree, Wm=10, W=24):
    """One step from the same state: the gradients the one-call step leaves in its workspace vs `.grad` on the autograd
    path.  This is the parity statement for the fused step (its projection kernels apply the pose transform themselves,
    so they are different instruction sequences from the autograd path's: equal to rounding, not bit for bit, on the GPU;
    trajectories then drift apart because Adam turns the rounding-noise gradients of the rotations — exactly zero in
    exact arithmetic for isotropic Gaussians — into +-lr steps, as it does from run to run in the reference)."""
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import FusedTrainer, _forward_backward_step, setup_training
    sc = syn_pointmap(3, Wm, Wm, W, W, seed=13)
    mk = lambda: setup_training(sc, dev, opt=OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True))
    a, b = mk(), mk()
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P")
    try:
        BinningPolicy.reset("exact")
        for st in (a, b):
            st.gaussians.active_sh_degree = degree
            if degree:  # something for the higher-order coefficients to differentiate through
                g = torch.Generator().manual_seed(3)
                with torch.no_grad():
                    st.gaussians._features_rest.copy_(0.05 * torch.randn(st.gaussians._features_rest.shape, generator=g).to(dev))
        loss_a = _forward_backward_step(a, True)
        ref = {n: getattr(a.gaussians, n).grad.detach().cpu() for n in names}
        tr = FusedTrainer(b, 200000)
        slot = torch.zeros(1, device=dev)
        tr.step(slot, defer_optimizer=True, verify_async=False)
        got = {n: v.cpu() for n, v in tr.gradients().items()}
        tr.close()
        cuda = torch.device(dev).type == "cuda"
        assert abs(float(slot) - float(loss_a)) <= (1e-6 if cuda else 0.0) * max(1.0, abs(float(loss_a)))
        for n in names:
            d = (got[n] - ref[n]).abs().max()
            if not cuda:
                assert float(d) == 0.0, (n, float(d))      # the emulator runs both paths with the same arithmetic
            else:
                # rounding-level relative to the tensor's own size.  The rotation gradient of an isotropic Gaussian is
                # exactly zero; what is computed is the cancellation residue of terms the size of the scale gradient,
                # so that is the yardstick for it.
                yard = float(ref["_scaling"].abs().max()) * 0.1 if n == "_rotation" else float(ref[n].abs().max())
                assert float(d) <= 1e-5 * yard, (n, float(d), yard)
        assert float(ref["_features_rest"].abs().max()) > 0 if degree else float(got["_features_rest"].abs().max()) == 0
    finally:
        BinningPolicy.reset("exact")


def check_run_ahead_crosses_sh_degree_step(dev, Wm=10, W=24):
    """Iteration 1000 raises the SH degree (reference train.py:148-149).  The one-call step covers every degree (it takes
    f_dc / f_rest as separate tensors), so it stays engaged across the change and must keep matching the autograd path —
    including the f_rest parameters, which only start receiving gradients at degree 1; the last iteration skips the
    optimizer."""
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import FusedTrainer, RunAhead, setup_training
    sc = syn_pointmap(3, Wm, Wm, W, W, seed=13)
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "P")
    cuda = torch.device(dev).type == "cuda"
    try:
        res = {}
        for fused in (False, True):
            st = generic_start(setup_training(sc, dev, opt=OptimizationParams(iterations=1003, pp_optimizer=True, optim_pose=True)))
            st.iteration = 996
            ra = RunAhead(st, window=2, fused_step=fused)
            assert (ra.trainer is not None) == fused
            for _ in range(7):          # iterations 997..1003
                ra.step()
            ema = ra.flush()
            assert st.iteration == 1003 and st.gaussians.active_sh_degree == 1 and FusedTrainer.supported(st)
            assert (ra.trainer is not None) == fused
            assert ema == ema and ema > 0  # finite
            for n in ("_xyz", "_features_rest"):
                s_ = st.gaussians.optimizer.state[getattr(st.gaussians, n)]["step"]
                assert s_ == 6, (n, s_)   # 7 iterations, the last one (== opt.iterations) without an optimizer step
            assert float(st.gaussians._features_rest.detach().abs().max()) > 0   # degree-1 coefficients are being trained
            assert float(st.gaussians._features_rest.detach()[:, 3:].abs().max()) == 0   # higher bands still untouched
            res[fused] = (ema, {n: getattr(st.gaussians, n).detach().cpu().clone() for n in names})
            BinningPolicy.reset("exact")
        bound("sh_degree_step/ema", abs(res[True][0] - res[False][0]) / max(1e-3, abs(res[False][0])), 2e-4 if cuda else 1e-6)
        for n in names:
            # f_rest included: its first Adam steps from zero are sign-like, but the gradients behind them are ordinary
            # (dL/dcolour times the SH basis), not rounding noise, so the two paths take the same steps
            bound("sh_degree_step/param" + n, rel_l2(res[True][1][n], res[False][1][n]), 2e-4 if cuda else 1e-5)
    finally:
        BinningPolicy.reset("exact")


def check_gated_off_tensor_keeps_moving(dev, Wm=10, W=24):
    """PerPointAdam's whole-tensor gate (reference scene/per_point_adam.py:62-69) freezes the MOMENTS of a tensor whose
    gradient is all zero but still applies the parameter step.  The one-call step remembers across launches that a
    gated-off tensor's first moment is all zero and then skips it without reading (adam.hip, MultiAdamArgs::live); that
    memory must be dropped as soon as the tensor has been updated.  Degree 0 (scan, skip, skip) -> degree 1 (f_rest
    trained) -> degree 0 again (f_rest gated off with live moments: it must keep moving exactly as on the autograd path)."""
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import RunAhead, setup_training
    sc = syn_pointmap(3, Wm, Wm, W, W, seed=17)
    cuda = torch.device(dev).type == "cuda"
    try:
        res = {}
        # False: autograd path through the compiled binding (AdamPlan keeps the library's `live` memory since ABI v7), True: the
        # one-call step (the trainer handle's `live`), "ctypes": autograd path through the ctypes binding, which passes NO memory —
        # every step reads the first moment again: the independent reference of the other two
        for fused in (False, True, "ctypes"):
            with _with_binding("ctypes" if fused == "ctypes" else "compiled"):
                st = generic_start(setup_training(sc, dev, opt=OptimizationParams(iterations=500, pp_optimizer=True, optim_pose=True)))
                ra = RunAhead(st, window=2, fused_step=(fused is True))
                g = st.gaussians
                track = []
                for degree in (0, 0, 0, 1, 1, 0, 0, 0):
                    g.active_sh_degree = degree
                    ra.step()
                    ra.flush()
                    track.append(g._features_rest.detach().cpu().clone())
                stt = g.optimizer.state[g._features_rest]
                res[fused] = (track, stt["exp_avg"].detach().cpu().clone(), stt["exp_avg_sq"].detach().cpu().clone())
                BinningPolicy.reset("exact")
        for k in range(8):
            a_, b_ = res[False][0][k], res["ctypes"][0][k]
            bound("gated_off/f_rest_track_compiled_vs_ctypes", float((a_ - b_).norm()) / float(b_.norm() + 1e-12) if float(b_.norm()) > 0 else float(a_.norm()),
                  2e-4 if cuda else 0.0)
        for fused in (False, True):
            track = res[fused][0]
            assert float(track[2].abs().max()) == 0                       # untouched while never updated
            assert float(track[4].abs().max()) > 0                        # trained at degree 1
            for a_, b_ in ((4, 5), (5, 6), (6, 7)):                       # gated off again: still moving on its momentum
                assert float((track[a_] - track[b_]).abs().max()) > 0, (fused, a_)
        for k in range(8):
            a_, b_ = res[True][0][k], res[False][0][k]
            bound("gated_off/f_rest_track", float((a_ - b_).norm()) / float(b_.norm() + 1e-12) if float(b_.norm()) > 0 else float(a_.norm()),
                  2e-4 if cuda else 1e-5)
        for i, what in ((1, "exp_avg"), (2, "exp_avg_sq")):
            bound("gated_off/f_rest_" + what, rel_l2(res[True][i], res[False][i]), 2e-5 if cuda else 1e-5)
        # the moments freeze while the tensor is gated off (both paths)
    finally:
        BinningPolicy.reset("exact")


def check_split_sh_equals_concatenated(dev, degree=2, P=300, W=64, H=48, seed=5):
    """GaussianRasterizer.forward(shs=f_dc, shs_rest=f_rest) must give what shs=cat(f_dc, f_rest) gives: same image, radii
    and gradients (reference scene/gaussian_model.py:129-132 concatenates; here the two tensors go in as they are)."""
    from instantsplat_amd.diff_gaussian_rasterization import GaussianRasterizer
    from instantsplat_amd.diff_gaussian_rasterization import GaussianRasterizationSettings
    from instantsplat_amd.synthetic import syn_blob
    from tests.util import settings_for
    dev = torch.device(dev)
    sc = syn_blob(P, W, H, seed=seed, scale_mean=0.05)
    settings = settings_for(sc.camera, degree, GaussianRasterizationSettings, torch.tensor([0.2, 0.5, 0.9]), device=dev)
    M = sc.shs.shape[1]
    outs = {}
    for split in (False, True):
        leaves = dict(means3D=sc.means3D, op=sc.opacity_logit, scaling=sc.scaling_logit, rot=sc.rotation)
        leaves = {k: v.clone().to(dev).requires_grad_(True) for k, v in leaves.items()}
        f_dc = sc.shs[:, :1].clone().to(dev).requires_grad_(True)
        f_rest = sc.shs[:, 1:].clone().to(dev).requires_grad_(True)
        means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
        kw = dict(means3D=leaves["means3D"], means2D=means2D, opacities=torch.sigmoid(leaves["op"]),
                  scales=torch.exp(leaves["scaling"]), rotations=leaves["rot"])
        r = GaussianRasterizer(settings)
        if split:
            img, radii = r(shs=f_dc, shs_rest=f_rest, **kw)
        else:
            img, radii = r(shs=torch.cat([f_dc, f_rest], 1), **kw)
        wgt = torch.linspace(0.2, 1.0, img.numel(), device=img.device).reshape(img.shape)
        (img * wgt).sum().backward()
        outs[split] = dict(img=img.detach().cpu(), radii=radii.cpu(), f_dc=f_dc.grad.cpu(), f_rest=f_rest.grad.cpu(),
                           means2D=means2D.grad.cpu(), **{k: v.grad.cpu() for k, v in leaves.items()})
    assert M == 16
    assert torch.equal(outs[True]["radii"], outs[False]["radii"])
    assert torch.equal(outs[True]["img"], outs[False]["img"])
    assert float(outs[False]["f_rest"].abs().max()) > 0
    cuda = torch.device(dev).type == "cuda"   # float atomics in the per-tile backward: order-dependent rounding on the GPU
    for k in outs[True]:
        if k in ("img", "radii"):
            continue
        a_, b_ = outs[True][k], outs[False][k]
        tol = 1e-4 if cuda else 0.0
        assert float((a_ - b_).abs().max()) <= tol * max(1.0, float(b_.abs().max())), k


def check_fused_synced_loop_equals_autograd_loop(dev, iters=5, force_overflow=False, Wm=12, W=32):
    """train_iteration(fused_step=True): reference loop shape (loss read back every iteration) on the one-call step,
    optimizer committed only after the read-back; must equal the autograd loop, also when every iteration overflows
    its instance buffers and is redone exactly."""
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import setup_training, train_iteration
    sc = syn_pointmap(3, Wm, Wm, W, W, seed=17)
    mk = lambda: generic_start(setup_training(sc, dev, opt=OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True)))
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P")
    cuda = torch.device(dev).type == "cuda"
    try:
        BinningPolicy.reset("exact")
        a = mk()
        la = [train_iteration(a) for _ in range(iters)]
        if force_overflow:
            BinningPolicy.slack, BinningPolicy.pad = 0.5, 0
        b = mk()
        lb = [train_iteration(b, fused_step=True) for _ in range(iters)]
        if not force_overflow:
            assert b._trainer is not None
        for x, y in zip(la, lb):
            bound("synced_one_call_vs_autograd/loss", abs(x - y) / max(1e-2, abs(x)), 1e-3 if cuda else 1e-6)   # MI355X: 6.5e-5
        for n in names:
            bound("synced_one_call_vs_autograd/param" + n, rel_l2(getattr(b.gaussians, n), getattr(a.gaussians, n)), 1e-3 if cuda else 1e-5)   # MI355X: <= 5.3e-5
    finally:
        BinningPolicy.slack, BinningPolicy.pad = 1.5, 16384
        BinningPolicy.reset("exact")


def check_dropin_node_housekeeping(dev, Wm=10, W=32, H=24):
    """What ABI v7 moved into the render node's own kernels, on the drop-in path (compiled binding):
      * `visibility_filter` is written by the projection kernel (no `radii > 0` launch): bool, equal to radii > 0;
      * the backward's accumulator buffer is allocated by the forward and cleared by the projection kernel — a SECOND backward of the
        same frame (retain_graph) must clear it itself and return the same gradients;
      * below its SH degree `f_rest.grad` is an alias of one persistent zero buffer: it must read as zeros, survive being modified in
        place (the next backward hands out zeros again), accumulate over two backward passes, and a no-grad render must not
        allocate anything for a backward."""
    from instantsplat_amd import _lib
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
    from instantsplat_amd.fused_ssim import fused_l1_ssim_loss
    from instantsplat_amd.gaussian_renderer import render
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import setup_training
    sc = syn_pointmap(2, Wm, Wm, W, H, seed=9)
    cuda = torch.device(dev).type == "cuda"
    try:
        with _with_binding("compiled"):
            st = generic_start(setup_training(sc, dev, opt=OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True)))
            g, cam = st.gaussians, st.cameras[0]
            names = ("_xyz", "_features_dc", "_opacity", "_scaling", "_rotation", "P")
            pkg = render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(cam.uid))
            vis = pkg["visibility_filter"]
            assert vis.dtype == torch.bool and vis.shape == pkg["radii"].shape and torch.equal(vis, pkg["radii"] > 0) and bool(vis.any())
            with torch.no_grad():
                pkg0 = render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(cam.uid))
            assert torch.equal(pkg0["visibility_filter"], vis) and pkg0["render"].grad_fn is None
            loss, _ = fused_l1_ssim_loss(pkg["render"].unsqueeze(0), st.gt_images[cam.uid].unsqueeze(0), 0.2)
            memsets = getattr(_lib.lib(), "mi355gs_emu_memset_calls", None)      # emulated library only: hipMemsetAsync calls so far
            n_memsets = memsets() if memsets else 0
            loss.backward(retain_graph=True)
            if memsets:   # the forward saw that a backward follows (asked OUTSIDE the custom function, where grad mode is on)
                assert memsets() == n_memsets, "the first backward of a frame cleared its accumulators with a memset"
            first = {n: getattr(g, n).grad.clone() for n in names}
            fr = g._features_rest.grad
            assert fr is not None and float(fr.abs().max()) == 0.0
            for n in names + ("_features_rest",):
                getattr(g, n).grad = None
            loss.backward()                                         # the same frame again: the accumulators were used once already
            if memsets:
                assert memsets() == n_memsets + 1, "a second backward of the same frame must clear the used accumulators itself"
            for n in names:
                bound("dropin_housekeeping/second_backward" + n, rel_l2(getattr(g, n).grad, first[n]), 2e-5 if cuda else 0.0)
            # ---- the shared zero gradient
            g._features_rest.grad.add_(1.0)                          # an in-place edit the version counter sees
            for n in names + ("_features_rest",):
                getattr(g, n).grad = None
            pkg = render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(cam.uid))
            fused_l1_ssim_loss(pkg["render"].unsqueeze(0), st.gt_images[cam.uid].unsqueeze(0), 0.2)[0].backward()
            assert float(g._features_rest.grad.abs().max()) == 0.0   # zeros again, not the polluted buffer
            pkg = render(st.cameras[1], g, st.pipe, st.background, camera_pose=g.get_RT(1))
            fused_l1_ssim_loss(pkg["render"].unsqueeze(0), st.gt_images[1].unsqueeze(0), 0.2)[0].backward()   # accumulates: zeros += zeros
            assert float(g._features_rest.grad.abs().max()) == 0.0
            # ---- two models of the same size in one process (teacher + student, a copy): each f_rest gets its OWN zero buffer — an
            # in-place edit of one model's .grad must not show up in the other's
            st2 = generic_start(setup_training(sc, dev, opt=OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True)))
            g2 = st2.gaussians
            for model in (g, g2):
                for n in names + ("_features_rest",):
                    getattr(model, n).grad = None
                pkg = render(cam, model, st.pipe, st.background, camera_pose=model.get_RT(cam.uid))
                fused_l1_ssim_loss(pkg["render"].unsqueeze(0), st.gt_images[cam.uid].unsqueeze(0), 0.2)[0].backward()
            assert g._features_rest.grad.data_ptr() != g2._features_rest.grad.data_ptr()
            g._features_rest.grad.add_(2.0)
            assert float(g2._features_rest.grad.abs().max()) == 0.0 and float(g._features_rest.grad.min()) == 2.0
            ext = _lib.compiled()
            was = ext.shared_zero_grad(False)
            try:
                for n in names + ("_features_rest",):
                    getattr(g, n).grad = None
                pkg = render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(cam.uid))
                fused_l1_ssim_loss(pkg["render"].unsqueeze(0), st.gt_images[cam.uid].unsqueeze(0), 0.2)[0].backward()
                a = g._features_rest.grad
                a.data.fill_(3.0)                                    # behind the version counter: harmless with a private tensor
                for n in names + ("_features_rest",):
                    getattr(g, n).grad = None
                pkg = render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(cam.uid))
                fused_l1_ssim_loss(pkg["render"].unsqueeze(0), st.gt_images[cam.uid].unsqueeze(0), 0.2)[0].backward()
                assert float(g._features_rest.grad.abs().max()) == 0.0
            finally:
                ext.shared_zero_grad(was)
    finally:
        BinningPolicy.reset("exact")


def check_synced_one_call_loop_can_be_left_and_reentered(dev, Wm=12, W=32):
    """The synced one-call loop runs the host half of iteration t + 1 under the device's work on iteration t.  Leaving it —
    for the autograd path, for RunAhead, for a look at the state — must take that half back: interleaving the three loops gives
    the trajectory of the plain reference-shaped loop, and st.iteration / the optimizer's step counts read right in between."""
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import RunAhead, cancel_prepared, release_trainer, setup_training, train_iteration
    sc = syn_pointmap(3, Wm, Wm, W, W, seed=23)
    mk = lambda: generic_start(setup_training(sc, dev, opt=OptimizationParams(iterations=14, pp_optimizer=True, optim_pose=True)))
    cuda = torch.device(dev).type == "cuda"
    try:
        BinningPolicy.reset("exact")
        a = mk()
        for _ in range(14):
            train_iteration(a)
        b = mk()
        for _ in range(3):
            train_iteration(b, fused_step=True)
        assert b._prepared is not None and b.iteration == 4            # the next iteration's host half has run ...
        cancel_prepared(b)
        assert b._prepared is None and b.iteration == 3                # ... and is taken back
        assert [b.gaussians.optimizer.state[p]["step"] for p in b._trainer.params] == [3] * 7
        train_iteration(b)                                             # autograd path (it cancels by itself)
        train_iteration(b, fused_step=True)
        train_iteration(b, fused_step=True)
        assert b.iteration == 7
        train_iteration(b)
        assert b.iteration == 7
        release_trainer(b)
        ra = RunAhead(b, window=3)
        for _ in range(3):
            ra.step()
        ra.flush()
        if ra.trainer is not None:
            ra.trainer.close()
        BinningPolicy.reset("exact")
        assert b.iteration == 10
        for _ in range(4):                                             # ... to the run's last iteration: no half iteration is left over
            train_iteration(b, fused_step=True)
        assert b.iteration == 14 and b._prepared is None
        assert [b.gaussians.optimizer.state[p]["step"] for p in b._trainer.params] == [13] * 7   # train.py:209: the last step is skipped
        release_trainer(b)
        for n in TRAIN_TENSORS:
            bound("leave_and_reenter/param" + n, rel_l2(getattr(b.gaussians, n), getattr(a.gaussians, n)), 1e-3 if cuda else 1e-5)
    finally:
        BinningPolicy.reset("exact")


def check_commit_gate_leaves_an_overflowed_step_uncommitted(dev, Wm=12, W=32):
    """mi355gs_trainer_step(do_optimizer_step = 1) enqueues the optimizer before the host has seen the frame's instance count:
    the Adam launch itself must write NOTHING — parameters, both moments — when the count exceeded the capacity of the instance
    buffers (device-side commit gate, csrc/trainer.hip), and must update everything when it did not."""
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
    from instantsplat_amd.gaussian_renderer import render
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import FusedTrainer, binning_hint, hint_key, setup_training
    sc = syn_pointmap(3, Wm, Wm, W, W, seed=19)
    st = generic_start(setup_training(sc, dev, opt=OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True)))
    g = st.gaussians
    try:
        BinningPolicy.reset("exact")
        with torch.no_grad():
            for cam in st.cameras:
                with binning_hint(hint_key(st, cam)):
                    render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(cam.uid))
        need = min(BinningPolicy.known[hint_key(st, c)] for c in st.cameras)
        words = torch.zeros(2, dtype=torch.int32, pin_memory=(torch.device(dev).type == "cuda"))
        loss_slot = words[0:1].view(torch.float32)
        sync = (lambda: torch.cuda.synchronize(dev)) if torch.device(dev).type == "cuda" else (lambda: None)

        def state():
            out = [getattr(g, n).detach().clone() for n in TRAIN_TENSORS]
            for n in TRAIN_TENSORS:
                s_ = g.optimizer.state[getattr(g, n)]
                out += [s_["exp_avg"].clone(), s_["exp_avg_sq"].clone()]
            return out
        big = FusedTrainer(st, 4 * max(BinningPolicy.known[hint_key(st, c)] for c in st.cameras) + 1024)
        before = state()
        big.step(loss_slot, verify_async=False, count_out=words[1:2])     # fits: commits (moments become non-zero)
        sync()
        assert int(words[1]) <= big.capacity
        after = state()
        for n, a, b in zip(TRAIN_TENSORS, before, after):   # (f_rest has no gradient at SH degree 0: its step is p - s * 0)
            assert torch.equal(a, b) == (n == "_features_rest"), n
        big.close()
        small = FusedTrainer(st, max(need // 2, 1))
        for _ in range(3):                                                # every view overflows this capacity
            small.step(loss_slot, verify_async=False, count_out=words[1:2])
            sync()
            assert int(words[1]) > small.capacity
        for a, b in zip(after, state()):
            assert torch.equal(a, b)                                      # bit for bit: nothing was written
        small.close()
    finally:
        BinningPolicy.reset("exact")


TRAIN_TENSORS = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P")


def _reference_loop_start(dev, run="loop"):
    """Initial state of a reference-driven training run recorded by tests/golden/make_golden.py (`loop_*` inputs; run "loop" =
    --pp_optimizer --optim_pose, 12 iterations; run "loopb" = plain Adam with fixed poses, 8 iterations)."""
    import os
    from instantsplat_amd.arguments import OptimizationParams, PipelineParams
    from instantsplat_amd.camera import Camera
    from instantsplat_amd.pose_utils import quadmultiply
    from instantsplat_amd.scene import GaussianModel, confidence_to_lr_modifiers
    from instantsplat_amd.train import TrainState
    import random
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))
    T = lambda k: torch.from_numpy(G[k])
    V, _, W, H, _ = [int(x) for x in G["loop_config"]]
    pp, optim_pose, iters = [int(x) for x in G[run + "_flags"]]
    cams = [Camera(v, T("loop_cam_w2c")[v], float(G["loop_cam_fov"][v, 0]), float(G["loop_cam_fov"][v, 1]), W, H) for v in range(V)]
    g = GaussianModel(3)
    g.create_from_pcd(T("loop_points_noisy"), T("loop_colors_noisy"), float(G["loop_extent"]), dev)
    with torch.no_grad():
        # the recorded run took its initial scales from the float64 k-d tree (the reference's distCUDA2 does not exist here)
        d2 = torch.clamp_min(knn_ref.dist2(T("loop_points_noisy")), 0.0000001)
        assert torch.allclose(g._scaling.detach().cpu(), torch.log(torch.sqrt(d2))[:, None].repeat(1, 3), rtol=0, atol=1e-5)
        g._scaling.copy_((torch.log(torch.sqrt(d2))[:, None].repeat(1, 3) + T("loop_init_scaling_delta")).to(dev))
        g._rotation.copy_(T("loop_init_rotation").to(dev))
    g.init_RT_seq(cams, dev)
    with torch.no_grad():
        P = g.P.detach().clone()
        P[:, :4] = quadmultiply(T("loop_pose_noise_q").to(dev), P[:, :4])
        P[:, 4:] += T("loop_pose_noise_t").to(dev)
    g.P = P.requires_grad_(True)
    with torch.no_grad():
        # Same initial BITS as the recorded run.  create_from_pcd on the device reproduces them to 1 ulp (RGB2SH divides by a
        # constant, which the GPU evaluates as a multiplication by its reciprocal), and 1 ulp is not harmless here: the
        # synthetic colours are clamped to [0, 1], so some channels are exactly 0, f_dc = RGB2SH(0) = -0.5/C0, and
        # C0 * f_dc + 0.5 lands on either side of the SH colour clamp (colour < 0 -> gradient masked) depending on that
        # last bit.  Found on MI355X in round 2: 17 of 324 f_dc elements got a zero gradient, stayed put in Adam's first
        # step (everyone else moves by lr) and the trajectory ended 2 % away in f_dc (tools/diag_step0.py).
        for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P"):
            ref0 = T(run + "_iter_params" + (n if n.startswith("_") else "_" + n))[0].to(dev)
            assert float((getattr(g, n).detach() - ref0).abs().max()) <= 1e-6, n
            getattr(g, n).copy_(ref0)
    opt = OptimizationParams(iterations=iters, pp_optimizer=bool(pp), optim_pose=bool(optim_pose))
    conf = confidence_to_lr_modifiers(T("loop_confidence").to(dev), scale=(1.0, 100.0))
    if pp:
        g.training_setup_pp(opt, conf)
    else:
        g.training_setup(opt)            # reference train.py:98-101
    cams = [c.to(dev) for c in cams]
    st = TrainState(g, cams, [T("loop_gt_images")[v].to(dev) for v in range(V)], torch.zeros(3, device=dev), opt, PipelineParams())
    st.rng = random.Random(0)
    return G, st, conf


def check_training_loop_matches_reference_function(dev, fused_step, run="loop"):
    """The device training loop vs a trajectory produced by the reference's OWN `training()` (train.py:87-230, executed by
    make_golden.py around the fp32 C oracle as the rasterizer operator): per-iteration losses, the view order, the LR
    schedule, the skipped optimizer step of the last iteration, the parameters after every iteration and at the end.  The
    start is non-degenerate (anisotropic scales, generic rotations: no structurally zero gradient for Adam to amplify) and
    bit-identical to the recorded run's (see _reference_loop_start for why the last bit of f_dc matters)."""
    import random
    from instantsplat_amd.train import FusedTrainer, train_iteration
    G, st, _ = _reference_loop_start(dev, run)
    iters = int(G[run + "_flags"][2])
    rng, stack, order = random.Random(0), [], []
    for _ in range(iters):          # reference train.py:152-157
        if not stack:
            stack = list(range(int(G["loop_config"][0])))
        order.append(stack.pop(rng.randint(0, len(stack) - 1)))
    assert order == list(G[run + "_view_uids"])
    if fused_step:
        assert FusedTrainer.supported(st)
    # Tolerances measured, not guessed: on MI355X (round 2, tools/diag_loop.py) the free-running loop ends within 6e-7
    # (relative L2, every tensor) of the reference's trajectory, op-by-op and one-call alike, the same as under the
    # emulator; the last iteration's loss is 2e-4 off on both (one pixel/Gaussian pair on the other side of alpha = 1/255).
    g = st.gaussians
    key = lambda kind, n: run + kind + (n if n.startswith("_") else "_" + n)
    for it in range(iters):
        l = float(train_iteration(st, fused_step=fused_step))
        assert abs(l - G[run + "_losses"][it]) <= 1e-3 * G[run + "_losses"][it], (it, l, G[run + "_losses"][it])
        if it + 1 < iters:   # the state the reference's run went into its next iteration with
            for n in TRAIN_TENSORS:
                a, b = getattr(g, n).detach().cpu(), torch.from_numpy(G[key("_iter_params", n)][it + 1])
                rel = float((a - b).norm() / (b.norm() + 1e-30))
                assert rel <= 2e-5, (it, n, rel)
    for n in TRAIN_TENSORS:
        a, b = getattr(g, n).detach().cpu(), torch.from_numpy(G[key("_final", n)])
        rel = float((a - b).norm() / (b.norm() + 1e-30))
        assert rel <= 2e-5, (n, rel)
    assert np.allclose([grp["lr"] for grp in g.optimizer.param_groups], G[run + "_final_lrs"], rtol=1e-12, atol=0)
    steps = [int(g.optimizer.state.get(grp["params"][0], {}).get("step", 0)) for grp in g.optimizer.param_groups]
    assert steps == [int(x) for x in G[run + "_final_steps"]], steps   # a fixed pose tensor never gets optimizer state


def check_teacher_forced_gradients_match_reference_function(dev, run="loop", fused_loss=True):
    """Teacher forcing: at every iteration of the reference's own `training()` run the recorded parameters
    are loaded, ONE forward + backward runs on the device path, and each tensor's gradient is compared with the gradient
    the reference run held right after its `loss.backward()` (train.py:177; recorded by tests/golden/make_golden.py through
    the `iter_end.record()` stand-in).  Unlike a trajectory this cannot hide or amplify anything: a kernel bug shows as a
    gradient mismatch in the iteration and tensor where it occurs.  Measured on MI355X: <= 8e-6 in every tensor/iteration."""
    from instantsplat_amd.fused_ssim import fused_l1_ssim_loss, fused_ssim
    from instantsplat_amd.gaussian_renderer import render
    G, st, _ = _reference_loop_start(dev, run)
    g = st.gaussians
    key = lambda kind, n: run + kind + (n if n.startswith("_") else "_" + n)
    pp, optim_pose, iters = [int(x) for x in G[run + "_flags"]]
    if not optim_pose:
        g.P.requires_grad_(False)
    for it in range(iters):
        with torch.no_grad():
            for n in TRAIN_TENSORS:
                getattr(g, n).copy_(torch.from_numpy(G[key("_iter_params", n)][it]).to(dev))
                getattr(g, n).grad = None
        uid = int(G[run + "_view_uids"][it])
        pkg = render(st.cameras[uid], g, st.pipe, st.background, camera_pose=g.get_RT(uid))
        image, gt = pkg["render"], st.gt_images[uid]
        if fused_loss == "train_py":
            # train.py:171-176 as written, on the drop-in modules (utils.loss_utils -> instantsplat_amd.loss_utils): the loss pair
            # and the recorded scalar expression of instantsplat_amd/lazy_loss.py
            from instantsplat_amd import lazy_loss
            from instantsplat_amd.loss_utils import l1_loss
            Ll1 = l1_loss(image, gt)
            ssim_value = fused_ssim(image.unsqueeze(0), gt.unsqueeze(0))
            assert isinstance(Ll1, lazy_loss.LazyScalar) and isinstance(ssim_value, lazy_loss.LazyScalar)
            loss = (1.0 - st.opt.lambda_dssim) * Ll1 + st.opt.lambda_dssim * (1.0 - ssim_value)
            assert isinstance(loss, lazy_loss.LazyScalar)
        elif fused_loss:
            loss, _ = fused_l1_ssim_loss(image.unsqueeze(0), gt.unsqueeze(0), st.opt.lambda_dssim)
        else:
            loss = (1.0 - st.opt.lambda_dssim) * (image - gt).abs().mean() \
                + st.opt.lambda_dssim * (1.0 - fused_ssim(image.unsqueeze(0), gt.unsqueeze(0)))
        loss.backward()
        ref_l = float(G[run + "_losses"][it])
        assert abs(float(loss.detach()) - ref_l) <= 1e-4 * ref_l, (it, float(loss.detach()), ref_l)
        for n in TRAIN_TENSORS:
            ref = torch.from_numpy(G[key("_iter_grads", n)][it]).double()
            got = getattr(g, n).grad
            if n == "P" and not optim_pose:
                assert got is None
                continue
            got = got.detach().cpu().double()
            if float(ref.abs().max()) == 0.0:      # f_rest below its SH degree: exactly zero in the reference, and here
                assert float(got.abs().max()) == 0.0, (it, n)
                continue
            rel = float((got - ref).norm() / ref.norm())
            assert rel <= 1e-4, (it, n, rel)
            # and element-wise where the reference gradient is exactly zero (masked SH clamp, culled Gaussians)
            assert float(got[ref == 0].abs().max() if bool((ref == 0).any()) else 0.0) <= 1e-7 * float(ref.abs().max()), (it, n)


def check_oracle_trainer_matches_reference_function(dev):
    """oracle/train_ref.CpuTrainer (the restated loop behind bench.py's cpu_baseline and the device-vs-oracle tests) vs the
    same reference-driven trajectory: same operator (the C oracle) on both sides, so the losses agree to rounding."""
    from oracle.train_ref import CpuTrainer
    G, st, conf = _reference_loop_start(dev)   # (the initial scales go through the 3-NN kernel once, then CPU only)
    g = st.gaussians
    params = dict(xyz=g._xyz, f_dc=g._features_dc, f_rest=g._features_rest, opacity=g._opacity, scaling=g._scaling,
                  rotation=g._rotation, pose=g.P)
    g.update_learning_rate(1)
    cpu = CpuTrainer(params, st.cameras, st.gt_images, conf, {grp["name"]: grp["lr"] for grp in g.optimizer.param_groups})
    for it in range(1, int(G["loop_config"][4]) + 1):
        g.update_learning_rate(it)
        for grp, dgrp in zip(cpu.opt.param_groups, g.optimizer.param_groups):
            grp["lr"] = dgrp["lr"]
        l = cpu.iteration()
        assert abs(l - G["loop_losses"][it - 1]) <= 5e-5 * G["loop_losses"][it - 1], (it, l, G["loop_losses"][it - 1])


def check_pose_tracking_matches_reference_function(dev):
    """pose_tracking.optimize_view_pose vs the reference's own `render_set_optimize` (render.py:99-186, executed by
    make_golden.py around the C oracle operator on the student of the recorded training run): the pose handed to every
    render of the 15 tracking iterations, the masked-L1 losses, the best pose and its final rendering."""
    import os
    from instantsplat_amd.arguments import PipelineParams
    from instantsplat_amd.camera import Camera
    from instantsplat_amd.scene import GaussianModel
    from instantsplat_amd import pose_tracking
    import instantsplat_amd.gaussian_renderer as gr
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))
    T = lambda k: torch.from_numpy(G[k])
    _, _, W, H, _ = [int(x) for x in G["loop_config"]]
    g = GaussianModel(3)
    for n in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        setattr(g, n, torch.nn.Parameter(T("loop_final" + n).clone().to(dev)))
    view = Camera(0, T("track_w2c_guess"), float(G["loop_cam_fov"][1, 0]), float(G["loop_cam_fov"][1, 1]), W, H, image=T("track_gt")).to(dev)
    poses = []
    real_render = gr.render

    def recording_render(cam, pc, pipe, bg, camera_pose=None, **k):
        poses.append(camera_pose.detach().cpu().clone())
        return real_render(cam, pc, pipe, bg, camera_pose=camera_pose, **k)

    pose_tracking.render = recording_render
    try:
        res = pose_tracking.render_set_optimize([view], g, PipelineParams(), torch.zeros(3, device=dev), num_iter=int(G["track_iters"]))[0]
    finally:
        pose_tracking.render = real_render
    ref_seq = T("track_pose_sequence")
    assert len(poses) == ref_seq.shape[0]
    bound("pose_tracking_ref/pose_sequence", (torch.stack(poses) - ref_seq).abs().max(), 2e-5)
    bound("pose_tracking_ref/initial_loss", abs(res["initial_loss"] - G["track_losses"][0]) / G["track_losses"][0], 1e-4)
    bound("pose_tracking_ref/best_loss", abs(res["best_loss"] - G["track_losses"].min()) / G["track_losses"].min(), 2e-4)
    bound("pose_tracking_ref/optimal_pose", (res["pose"].cpu() - T("track_optimal_pose")).abs().max(), 2e-5)
    bound("pose_tracking_ref/final_render", (res["render"].cpu() - T("track_final_render")).abs().max(), 2e-4)
    for t in (g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation):
        assert not t.requires_grad


def check_capture_matches_reference_class(dev):
    """GaussianModel.capture() after the recorded 12-iteration run vs the tuple the reference's own class returned after
    the same run (tests/golden `capture_*`, scene/gaussian_model.py:65-80): 13 entries in the reference's order, parameter
    values, densification placeholders, and the optimizer state_dict inside it (group keys / names / lrs, per-parameter
    step, exp_avg, exp_avg_sq) — what makes chkpnt<iteration>.pth interchangeable."""
    from instantsplat_amd.train import train_iteration
    G, st, _ = _reference_loop_start(dev, "loop")
    for _ in range(int(G["loop_flags"][2])):
        train_iteration(st)
    cap = st.gaussians.capture()
    assert len(cap) == int(G["capture_len"]) == 13
    assert cap[0] == int(G["capture_active_sh_degree"])
    for i, name in zip(range(1, 7), ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")):
        ref = torch.from_numpy(G["capture" + name])
        assert tuple(cap[i].shape) == tuple(ref.shape) and isinstance(cap[i], torch.nn.Parameter), name
        bound("capture/param" + name, rel_l2(cap[i], ref), 2e-5)
    for i, name in ((7, "max_radii2D"), (8, "xyz_gradient_accum"), (9, "denom")):
        ref = torch.from_numpy(G["capture_" + name])
        assert tuple(cap[i].shape) == tuple(ref.shape) and float(cap[i].abs().max()) == float(ref.abs().max()) == 0.0, name
    sd = cap[10]
    assert sorted(sd["param_groups"][0].keys()) == list(G["capture_opt_group_keys"])
    assert [g["name"] for g in sd["param_groups"]] == list(G["capture_opt_group_names"])
    assert [g["params"][0] for g in sd["param_groups"]] == list(G["capture_opt_group_params"])
    assert np.allclose([g["lr"] for g in sd["param_groups"]], G["capture_opt_group_lrs"], rtol=1e-12, atol=0)
    assert sorted(sd["state"].keys()) == list(G["capture_opt_state_ids"])
    for k, s_ in sd["state"].items():
        assert sorted(s_.keys()) == list(G[f"capture_opt_state{k}_keys"])
        assert int(s_["step"]) == int(G[f"capture_opt_state{k}_step"])
        for m in ("exp_avg", "exp_avg_sq"):
            ref = torch.from_numpy(G[f"capture_opt_state{k}_{m}"])
            if float(ref.abs().max()) == 0.0:
                assert float(s_[m].abs().max()) == 0.0, (k, m)
            else:
                bound(f"capture/opt_state_{m}[{k}]", rel_l2(s_[m], ref), 5e-5)
    assert abs(float(cap[11]) - float(G["capture_spatial_lr_scale"])) <= 1e-6 * float(G["capture_spatial_lr_scale"])
    bound("capture/P", rel_l2(cap[12], torch.from_numpy(G["capture_P"])), 1e-6)


def check_checkpoint_save_and_resume(dev, tmp_path, Wm=10, W=24):
    """training(model_path=..., saving_iterations, checkpoint_iterations, start_checkpoint): the files the reference writes
    (train.py:107-110,220-227) appear with their formats, and a run resumed from chkpnt<k>.pth starts at iteration k with the
    parameters, poses, optimizer moments / step counts and learning-rate schedule the interrupted run had there."""
    import os
    from instantsplat_amd.io_formats import load_gaussian_ply
    from instantsplat_amd.scene import GaussianModel
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import training
    sc = syn_pointmap(3, Wm, Wm, W, W, seed=19)
    mp = str(tmp_path)
    r = training(sc, dev, iterations=9, model_path=mp, saving_iterations=(6, 9), checkpoint_iterations=(6,))
    g = r["state"].gaussians
    for it in (6, 9):
        assert os.path.exists(os.path.join(mp, "point_cloud", f"iteration_{it}", "point_cloud.ply"))
        org, optd = (np.load(os.path.join(mp, "pose", f"ours_{it}", n + ".npy")) for n in ("pose_org", "pose_optimized"))
        assert org.shape == optd.shape == (3, 4, 4) and float(np.abs(org - optd).max()) > 0
    ply = load_gaussian_ply(os.path.join(mp, "point_cloud", "iteration_9", "point_cloud.ply"), 3, "cpu")
    for k, v in ply.items():
        assert torch.equal(v, getattr(g, k).detach().cpu()), k
    m2 = GaussianModel(3)
    m2.load_ply(os.path.join(mp, "point_cloud", "iteration_9", "point_cloud.ply"), device=dev)
    assert m2.active_sh_degree == 3 and torch.equal(m2._features_rest.detach().cpu(), g._features_rest.detach().cpu())
    ck_params, ck_iter = torch.load(os.path.join(mp, "chkpnt6.pth"), map_location="cpu", weights_only=False)
    assert ck_iter == 6 and len(ck_params) == 13
    # resume: zero further iterations -> the restored state itself
    r2 = training(sc, dev, iterations=6, start_checkpoint=os.path.join(mp, "chkpnt6.pth"))
    g2 = r2["state"].gaussians
    assert r2["state"].iteration == 6
    for i, name in zip(range(1, 7), ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")):
        assert torch.equal(getattr(g2, name).detach().cpu(), ck_params[i].detach().cpu()), name
    assert torch.equal(g2.P.detach().cpu(), ck_params[12].detach().cpu())
    sd = g2.optimizer.state_dict()
    for k, s_ in ck_params[10]["state"].items():
        assert int(sd["state"][k]["step"]) == int(s_["step"]) == 6
        assert torch.equal(sd["state"][k]["exp_avg"].cpu(), s_["exp_avg"].cpu()) and torch.equal(sd["state"][k]["exp_avg_sq"].cpu(), s_["exp_avg_sq"].cpu())
    assert g2.optimizer.param_groups[0].get("per_point_lr") is not None     # the per-point multiplier survives the resume
    # and it keeps training from there: iterations 7..9 run, the loss stays finite, parameters move
    r3 = training(sc, dev, iterations=9, start_checkpoint=os.path.join(mp, "chkpnt6.pth"))
    assert r3["state"].iteration == 9 and r3["last_loss"] == r3["last_loss"]
    assert float((r3["state"].gaussians._xyz.detach().cpu() - ck_params[1].detach().cpu()).abs().max()) > 0
    steps = [int(r3["state"].gaussians.optimizer.state[p]["step"]) for p in (r3["state"].gaussians._xyz, r3["state"].gaussians.P)]
    assert steps == [8, 8]   # 6 restored + iterations 7 and 8 (the last iteration, 9, skips the optimizer)


# ---- compiled PyTorch binding (instantsplat_amd/csrc_torch/binding.cpp) ---------------------------------------------
def _with_binding(name):
    import contextlib
    from instantsplat_amd import _lib

    if name == "compiled" and torch.version.hip is None:
        import pytest
        pytest.skip("the compiled binding needs a ROCm build of PyTorch (tests/conftest.py falls back to ctypes)")

    @contextlib.contextmanager
    def cm():
        prev, _lib.BINDING = _lib.BINDING, name
        try:
            yield
        finally:
            _lib.BINDING = prev
    return cm()


def check_compiled_binding_equals_ctypes(dev, iters=5, Wm=12, W=40, H=32):
    """The drop-in loop (train_iteration: render -> loss -> backward -> loss.item() -> PerPointAdam.step) through the compiled
    binding vs through the ctypes / Python autograd.Function binding: the same C-ABI calls with the same arguments, so under
    the emulator (deterministic atomics) every loss and parameter is bit-identical; on the GPU to float-atomic order.  Also
    the statement that the compiled Adam took the backward's gate flags for all seven tensors (since ABI v8 the pose TABLE's
    gradient is written by the backward too, PoseRowFn; before, autograd scattered it from one row) — and that a run in
    which f_rest is gated off (degree 0), trained (degree 1) and gated off again keeps matching."""
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import setup_training, train_iteration
    sc = syn_pointmap(3, Wm, Wm, W, H, seed=21)
    cuda = torch.device(dev).type == "cuda"
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P")
    res = {}
    try:
        for binding in ("ctypes", "compiled"):
            with _with_binding(binding):
                st = generic_start(setup_training(sc, dev, opt=OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True)))
                losses = []
                for degree in [0] * iters + [1, 1, 0, 0]:
                    st.gaussians.active_sh_degree = degree
                    losses.append(train_iteration(st))
                g = st.gaussians
                res[binding] = (losses, {n: getattr(g, n).detach().cpu().clone() for n in names},
                                {n: g.optimizer.state[getattr(g, n)]["exp_avg_sq"].detach().cpu().clone() for n in names})
                if binding == "compiled":
                    plans = [b["compiled"] for pl in g.optimizer._plans.values() for b in pl["batches"] if "compiled" in b]
                    assert plans and all(p.last_used_gates == 7 for p in plans), [(p.last_used_gates, p.last_gate_note) for p in plans]
            BinningPolicy.reset("exact")
        for a_, b_ in zip(res["ctypes"][0], res["compiled"][0]):
            bound("compiled_vs_ctypes/loss", abs(a_ - b_) / max(abs(a_), 1e-6), 1e-3 if cuda else 0.0)   # MI355X, two runs of one binding: 1.4e-4
        for n in names:
            for k, what in ((1, "param"), (2, "exp_avg_sq")):
                # GPU: float-atomic order only.  f_rest takes its first Adam steps in this run (degree 1), and a first step is
                # lr * sign(g) wherever |g| >> eps: ONE element of 311k whose tiny gradient changes sign between two runs moves
                # the relative L2 of the whole tensor to 2.4e-3 (measured) — the bound for it is that of a handful of such flips
                lim = (2e-2 if n == "_features_rest" else 2e-4) if k == 1 else 2e-3
                bound("compiled_vs_ctypes/%s%s" % (what, n), rel_l2(res["compiled"][k][n], res["ctypes"][k][n]), lim if cuda else 0.0)
    finally:
        BinningPolicy.reset("exact")


def check_compiled_gate_flags_are_sound(dev, Wm=10, W=32, H=24):
    """The compiled Adam may take the backward's gate flags only for gradients that are provably the tensors that backward
    wrote.  Everything else must fall back to summing the gradient: a gradient accumulated over two backward passes, a
    gradient modified in place (clipping), a gradient replaced by a copy — and in every case the update must equal the
    ctypes binding's (which always sums)."""
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
    from instantsplat_amd.fused_ssim import fused_l1_ssim_loss
    from instantsplat_amd.gaussian_renderer import render
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import setup_training
    sc = syn_pointmap(2, Wm, Wm, W, H, seed=4)
    cuda = torch.device(dev).type == "cuda"
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P")

    def one_backward(st, cam):
        g = st.gaussians
        img = render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(cam.uid))["render"]
        loss, _ = fused_l1_ssim_loss(img.unsqueeze(0), st.gt_images[cam.uid].unsqueeze(0), st.opt.lambda_dssim)
        loss.backward()

    def scenario(kind):
        st = generic_start(setup_training(sc, dev, opt=OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True)))
        g = st.gaussians
        g.update_learning_rate(1)
        one_backward(st, st.cameras[0])
        if kind == "accumulated":
            one_backward(st, st.cameras[1])
        elif kind == "clipped":
            with torch.no_grad():
                g._xyz.grad.mul_(0.5)
        elif kind == "replaced":
            g._opacity.grad = g._opacity.grad.clone()
        elif kind == "zeroed":   # the flag says "non-zero" for a gradient that has since been zeroed in place
            with torch.no_grad():
                g._scaling.grad.zero_()
        elif kind == "zeroed_behind_the_version_counter":
            # `.data` edits do not bump `_version()`: the flag would be stale.  The documented contract (optim.PerPointAdam) is
            # that code editing gradients this way switches the shortcut off — then the step sums the gradients itself.
            g._scaling.grad.data.zero_()
            g.optimizer.use_backward_gates = False
        g.optimizer.step()
        used = None
        plans = [b.get("compiled") for pl in getattr(g.optimizer, "_plans", {}).values() for b in pl["batches"]]
        if plans and plans[0] is not None:
            used = plans[0].last_used_gates
        return {n: getattr(g, n).detach().cpu().clone() for n in names}, {
            n: g.optimizer.state[getattr(g, n)]["exp_avg_sq"].detach().cpu().clone() for n in names}, used

    try:
        # (seven with the pose table, whose gradient the backward writes too since ABI v8 — get_RT is a node of the binding)
        expect = {"fresh": 7, "accumulated": 0, "clipped": 6, "replaced": 6, "zeroed": 6, "zeroed_behind_the_version_counter": 0}
        for kind, n_flags in expect.items():
            with _with_binding("ctypes"):
                pa, va, _ = scenario(kind)
            with _with_binding("compiled"):
                pb, vb, used = scenario(kind)
            assert used == n_flags, (kind, used)
            for n in names:
                bound("compiled_gates/%s/param%s" % (kind, n), rel_l2(pb[n], pa[n]), 2e-5 if cuda else 0.0)   # MI355X: <= 8e-8
                bound("compiled_gates/%s/exp_avg_sq%s" % (kind, n), rel_l2(vb[n], va[n]), 2e-4 if cuda else 0.0)
            if kind.startswith("zeroed"):   # Adam's whole-tensor gate: a zeroed gradient must leave the second moment untouched
                assert float(vb["_scaling"].abs().max()) == 0.0
    finally:
        BinningPolicy.reset("exact")


def check_operator_bindings_agree(dev):
    """GaussianRasterizer.forward / backward and fused_ssim through the compiled nodes vs through the ctypes / Python
    autograd.Function nodes — all four input variants of the operator (SH or precomputed colours, scale/rotation or
    precomputed covariance): the same C-ABI calls, so identical under the emulator and equal to float-atomic order on the GPU."""
    from instantsplat_amd.fused_ssim import fused_ssim
    from tests.util import relerr, run_blob_case
    cuda = torch.device(dev).type == "cuda"
    for precomp_color, precomp_cov, deg in ((False, False, 2), (True, False, 0), (False, True, 1), (True, True, 0)):
        res = {}
        for binding in ("ctypes", "compiled"):
            with _with_binding(binding):
                res[binding] = run_blob_case(dev, 700, 80, 48, deg, precomp_color=precomp_color, precomp_cov=precomp_cov)["dut"]
        a, b = res["ctypes"], res["compiled"]
        tag = "bindings/color%d_cov%d/" % (precomp_color, precomp_cov)
        assert bool((a["radii"] == b["radii"]).all())
        bound(tag + "image", float((a["color"] - b["color"]).abs().max()), 1e-6 if cuda else 0.0)
        assert set(a["grads"]) == set(b["grads"])
        for k in a["grads"]:
            bound(tag + "grad_" + k, relerr(b["grads"][k], a["grads"][k]), 2e-6 if cuda else 0.0)
    g = torch.Generator().manual_seed(2)
    x = torch.rand(1, 3, 40, 56, generator=g).to(dev)
    y = torch.rand(1, 3, 40, 56, generator=g).to(dev)
    for padding in ("same", "valid"):
        vals = {}
        for binding in ("ctypes", "compiled"):
            with _with_binding(binding):
                xi = x.clone().requires_grad_(True)
                v = fused_ssim(xi, y, padding=padding)
                (3.0 * v).backward()
                vals[binding] = (float(v.detach()), xi.grad.detach().cpu().clone())
        bound("bindings/ssim_%s/value" % padding, abs(vals["ctypes"][0] - vals["compiled"][0]), 1e-7 if cuda else 0.0)
        bound("bindings/ssim_%s/grad" % padding, relerr(vals["compiled"][1], vals["ctypes"][1]), 1e-6 if cuda else 0.0)


def check_trainer_keeps_its_unit_length_knob(dev, Wm=14, W=48, H=32):
    """A trainer handle lays its workspace out for the unit-length knob (mi355gs_tune_min_units) as it stood at create; later
    steps must keep sizing and launching with that value even if the process-wide knob has been changed since (ADVICE r2:
    the step used to re-read the knob, so a change between create and step moved the unit table past its allocation)."""
    from instantsplat_amd import _lib
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import setup_training, train_iteration
    L = _lib.lib()
    sc = syn_pointmap(3, Wm, Wm, W, H, seed=13)
    names = ("_xyz", "_features_dc", "_opacity", "_scaling", "_rotation", "P")
    old = L.mi355gs_tune_min_units(0)
    cuda = torch.device(dev).type == "cuda"
    res = {}
    try:
        for change in (False, True):
            L.mi355gs_tune_min_units(2)          # long units on this small scene: the multi-chunk layout
            st = generic_start(setup_training(sc, dev, opt=OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True)))
            losses = [train_iteration(st, fused_step=True)]
            assert getattr(st, "_trainer", None) is not None
            if change:
                L.mi355gs_tune_min_units(old)    # the handle must not notice
            losses += [train_iteration(st, fused_step=True) for _ in range(4)]
            res[change] = (losses, {n: getattr(st.gaussians, n).detach().cpu().clone() for n in names})
            from instantsplat_amd.train import release_trainer
            release_trainer(st)
            BinningPolicy.reset("exact")
        for a_, b_ in zip(res[False][0], res[True][0]):
            bound("trainer_knob/loss", abs(a_ - b_) / max(abs(a_), 1e-6), 1e-3 if cuda else 0.0)
        for n in names:
            bound("trainer_knob/param" + n, rel_l2(res[True][1][n], res[False][1][n]), 2e-4 if cuda else 0.0)
    finally:
        L.mi355gs_tune_min_units(old)
        BinningPolicy.reset("exact")


def check_pose_row_node(dev, Wm=10, W=32, H=24):
    """GaussianModel.get_RT through the compiled binding (PoseRowFn, ABI v8 pose_rows / pose_row) against plain `P[idx]`:
      * same values, same memory as the table's row;
      * after render -> loss -> backward, P.grad is the [views, 7] table with the gradient in row idx and EXACT zeros elsewhere,
        equal to what autograd's select-backward gives for the same frame — written by the render node's last kernel (no
        fill, no copy: the table arrives as an alias of the memory that kernel wrote);
      * the optimizer then finds the table among "the gradients this backward wrote" (gated by the backward's flag);
      * a row that is ALSO used by something else (its gradient arrives as a sum), a row rendered through the op-by-op path, and
        two rows alive at once (the render node must not mistake one for the other) all give P[idx]'s gradients;
      * negative and out-of-range indices behave like indexing."""
    from instantsplat_amd import _lib
    from instantsplat_amd.arguments import OptimizationParams
    from instantsplat_amd.fused_ssim import fused_l1_ssim_loss
    from instantsplat_amd.gaussian_renderer import render
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import setup_training
    import pytest
    sc = syn_pointmap(3, Wm, Wm, W, H, seed=4)
    cuda = torch.device(dev).type == "cuda"
    with _with_binding("compiled"):
        ext = _lib.compiled()
        st = generic_start(setup_training(sc, dev, opt=OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True)))
        g = st.gaussians
        names = ("_xyz", "_features_dc", "_opacity", "_scaling", "_rotation", "P")

        def clear():
            for n in names + ("_features_rest",):
                getattr(g, n).grad = None

        def frame(cam, pose, extra=None):
            clear()
            pkg = render(cam, g, st.pipe, st.background, camera_pose=pose)
            loss, _ = fused_l1_ssim_loss(pkg["render"].unsqueeze(0), st.gt_images[cam.uid].unsqueeze(0), 0.2)
            if extra is not None:
                loss = loss + extra
            loss.backward()
            return {n: getattr(g, n).grad.clone() for n in names}

        cam = st.cameras[1]
        row = g.get_RT(1)
        assert row.grad_fn is not None and "PoseRow" in row.grad_fn.name(), row.grad_fn.name()
        assert torch.equal(row, g.P[1]) and row.data_ptr() == g.P[1].data_ptr()
        assert torch.equal(g.get_RT(-1), g.P[-1])
        with pytest.raises((RuntimeError, IndexError)):
            g.get_RT(3)
        ref = frame(cam, g.P[1])                       # autograd's own selection: zeros + copy
        memsets = getattr(_lib.lib(), "mi355gs_emu_memset_calls", None)
        got = frame(cam, g.get_RT(1))
        assert g.P.grad.shape == g.P.shape and g.P.grad.is_contiguous()
        assert float(g.P.grad[0].abs().max()) == 0.0 and float(g.P.grad[2].abs().max()) == 0.0 and float(g.P.grad[1].abs().max()) > 0.0
        for n in names:
            bound("pose_row/" + n, rel_l2(got[n], ref[n]), 2e-5 if cuda else 0.0)
        # the optimizer's fast path takes all seven gradients from the backward's record (pose table included)
        opt = g.optimizer
        opt.step()
        plan = opt._fast[2] if getattr(opt, "_fast", None) else None
        if plan is not None:
            assert plan.last_used_gates == 7, (plan.last_used_gates, plan.last_gate_note)
        # ---- the general path: the row's gradient arrives as a sum
        row = g.get_RT(1)
        got = frame(cam, row, extra=(row * row).sum() * 0.5)
        clear()
        pr = g.P[1]
        ref = frame(cam, pr, extra=(pr * pr).sum() * 0.5)
        for n in names:
            bound("pose_row_sum/" + n, rel_l2(got[n], ref[n]), 2e-5 if cuda else 1e-6)
        # ---- two rows alive at once: the render of camera 0 gets row 0, while row 2 was handed out last
        r0, r2 = g.get_RT(0), g.get_RT(2)
        got = frame(st.cameras[0], r0)
        ref = frame(st.cameras[0], g.P[0])
        for n in names:
            bound("pose_row_two/" + n, rel_l2(got[n], ref[n]), 2e-5 if cuda else 0.0)
        assert float(g.P.grad[2].abs().max()) == 0.0
        del r2


def check_loss_utils_against_the_references_own(dev):
    """instantsplat_amd.loss_utils (the drop-in for the reference's utils/loss_utils.py) against vectors its functions produced
    (tests/golden/make_golden_loss_utils.py): l1_loss through both bindings — value to a relative 3e-7 (per-workgroup float
    sums finished in double against PyTorch's float tree), gradient BIT for bit (sgn(a - b) * g / n, zero at exact ties) —,
    l2_loss, ssim for the fused case (11 x 11, mean) and the cases that stay PyTorch (other window, per-image mean), l1_loss_mask."""
    import os
    import numpy as np
    from instantsplat_amd import loss_utils
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_utils_vectors.npz"))
    T = lambda k: torch.from_numpy(G[k]).to(dev)
    cuda = torch.device(dev).type == "cuda"
    from instantsplat_amd import lazy_loss
    for binding, lazy in (("compiled", True), ("compiled", False), ("ctypes", True), ("ctypes", False)):
        lazy_was, lazy_loss.ENABLED = lazy_loss.ENABLED, lazy
        with _with_binding(binding):
            for k in range(5):
                a, b = T(f"l1_{k}_a").requires_grad_(True), T(f"l1_{k}_b")
                v = loss_utils.l1_loss(a, b)
                # an image-shaped pair goes through the loss pair (lazy_loss.py: L1 and SSIM in one pass, one node for the
                # whole scalar expression), everything else — and everything with the mechanism off — through the L1 node
                paired = lazy and a.dim() in (3, 4)
                assert isinstance(v, lazy_loss.LazyScalar) == paired, (binding, lazy, k, type(v), a.shape)
                assert v.dim() == 0 and v.grad_fn is not None and ("LossAffine" if paired else "L1Loss") in v.grad_fn.name(), v.grad_fn
                (v * 1.7).backward()
                bound("loss_utils/l1_value", abs(float(v) - float(G[f"l1_{k}_value"])) / float(G[f"l1_{k}_value"]), 3e-7)
                assert torch.equal(a.grad.cpu(), torch.from_numpy(G[f"l1_{k}_grad"])), (binding, k)
                bound("loss_utils/l2_value", abs(float(loss_utils.l2_loss(a.detach(), b)) - float(G[f"l2_{k}_value"])) / float(G[f"l2_{k}_value"]), 3e-7)
            # non-contiguous input, and a gt that wants a gradient (the kernel gives none: PyTorch's expression takes over)
            a, b = T("l1_1_a"), T("l1_1_b")
            at = a.transpose(1, 2).contiguous().transpose(1, 2).requires_grad_(True)
            assert not at.is_contiguous()
            bound("loss_utils/l1_value_strided", abs(float(loss_utils.l1_loss(at, b)) - float(G["l1_1_value"])) / float(G["l1_1_value"]), 3e-7)
            bg = b.clone().requires_grad_(True)
            loss_utils.l1_loss(a, bg).backward()
            assert bg.grad is not None and float(bg.grad.abs().sum()) > 0
        lazy_loss.ENABLED = lazy_was
        lazy_loss.forget()
    x, y, mask = T("ssim_x"), T("ssim_y"), T("ssim_mask")
    bound("loss_utils/ssim_11", abs(float(loss_utils.ssim(x, y)) - float(G["ssim_11"])), 2e-6 if cuda else 1e-6)
    bound("loss_utils/ssim_3d", abs(float(loss_utils.ssim(x[0], y[0])) - float(G["ssim_3d"])), 2e-6 if cuda else 1e-6)
    bound("loss_utils/ssim_7", abs(float(loss_utils.ssim(x, y, window_size=7)) - float(G["ssim_7"])), 2e-6)
    # a second image that wants a gradient: the fused kernel differentiates img1 only, so the reference's conv2d expression takes over
    # (same value, and img2 gets ITS gradient); under no_grad the fused kernel is taken again
    yg = y.clone().requires_grad_(True)
    v = loss_utils.ssim(x, yg)
    v.backward()
    bound("loss_utils/ssim_11_grad_img2_path", abs(float(v.detach()) - float(G["ssim_11"])), 2e-6)
    assert yg.grad is not None and float(yg.grad.abs().sum()) > 0
    with torch.no_grad():
        bound("loss_utils/ssim_11_no_grad", abs(float(loss_utils.ssim(x, yg)) - float(G["ssim_11"])), 2e-6 if cuda else 1e-6)
    bound("loss_utils/ssim_per_image", float((loss_utils.ssim(x, y, size_average=False).cpu() - torch.from_numpy(G["ssim_11_per_image"])).abs().max()), 2e-6)
    bound("loss_utils/l1_mask", abs(float(loss_utils.l1_loss_mask(x, y, mask)) - float(G["l1_mask"])), 1e-6)
    # every name the reference's module defines is here (render.py:30 imports ssim_loss_mask next to the others), with its results
    assert all(callable(getattr(loss_utils, str(n), None)) for n in G["names"]), [str(n) for n in G["names"] if not hasattr(loss_utils, str(n))]
    assert torch.equal(loss_utils.gaussian(11, 1.5), torch.from_numpy(G["gaussian_11"])) and torch.equal(loss_utils.create_window(7, 3), torch.from_numpy(G["window_7_3"]))
    bound("loss_utils/_ssim_7", abs(float(loss_utils._ssim(x, y, loss_utils.create_window(7, 3).to(dev), 7, 3, True)) - float(G["ssim_core_7"])), 2e-6)
    bound("loss_utils/ssim_mask_11", abs(float(loss_utils.ssim_loss_mask(x, y, mask)) - float(G["ssim_mask_11"])), 2e-6 if cuda else 1e-6)
    bound("loss_utils/ssim_mask_7_per_image", float((loss_utils.ssim_loss_mask(x, y, mask, window_size=7, size_average=False).cpu()
                                                       - torch.from_numpy(G["ssim_mask_7_per_image"])).abs().max()), 2e-6)


def check_render_only_forward(dev, Wm=20, W=80, H=48):
    """A render no backward can follow (torch.no_grad(), or nothing requiring a gradient) takes the render-only stage 2
    (mi355gs_raster_forward_render_only: keys + lists only in `binning`, no boundary records / hit masks / unit table / quadrant
    maxima stored).  Image, radii and the per-pixel state (final_T, n_contrib -> frame statistics) must be bit-identical to the
    training instantiation's — through render() on both bindings, through the operator, and with the switch off."""
    import instantsplat_amd.diff_gaussian_rasterization as dgr
    from instantsplat_amd import _lib
    from instantsplat_amd.gaussian_renderer import render
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import setup_training
    from tests.util import run_blob_case
    L = _lib.lib()
    sc = syn_pointmap(3, Wm, Wm, W, H, seed=9)
    st = generic_start(setup_training(sc, dev))
    g, cam = st.gaussians, st.cameras[1]
    pose = g.get_RT(cam.uid)

    def frame(no_grad):
        """render() through the Python node with the frame's buffers kept: image, radii, stats, binning bytes"""
        dgr.keep_last_frame(True)
        try:
            if no_grad:
                with torch.no_grad():
                    pkg = render(cam, g, st.pipe, st.background, camera_pose=pose)
            else:
                pkg = render(cam, g, st.pipe, st.background, camera_pose=pose)
            stats = dgr.last_frame_stats()
            lf = dgr._LAST_FRAME
            return (pkg["render"].detach().cpu().clone(), pkg["radii"].cpu().clone(), stats, int(lf["binning"].numel()), int(lf["capacity"]),
                    lf["tiles"].cpu().clone())
        finally:
            dgr.keep_last_frame(False)

    img_t, rad_t, stats_t, bytes_t, cap_t, tiles_t = frame(no_grad=False)
    img_r, rad_r, stats_r, bytes_r, cap_r, tiles_r = frame(no_grad=True)
    assert cap_t == cap_r and stats_t == stats_r and stats_t[0] > 0
    assert bytes_t == int(L.mi355gs_raster_binning_bytes(cap_t, W, H))
    assert bytes_r == max(int(L.mi355gs_raster_binning_bytes_render_only(cap_r, W, H)), 1) < bytes_t
    assert torch.equal(img_t, img_r) and torch.equal(rad_t, rad_r)
    # the per-pixel state the two instantiations share (final_T, n_contrib: include/mi355gs.h `tiles`) — the words in front of
    # the backward-only tables are the same bits
    npix_words = W * H
    nt = ((W + 15) // 16) * ((H + 15) // 16)
    al = lambda n: (n + 255) // 256 * 256
    off_final_T = 2 * al(nt * 4) + al((nt + 1) * 4)   # behind count, cursor and start (csrc/common.h TilesLayout)
    a = tiles_t[off_final_T: off_final_T + 8 * npix_words]
    b = tiles_r[off_final_T: off_final_T + 8 * npix_words]
    assert torch.equal(a, b)
    # render() on the compiled binding, the switch on and off, under no_grad and with requires_grad inputs
    ext = _lib.compiled()
    if ext is not None:
        with torch.no_grad():
            i1 = render(cam, g, st.pipe, st.background, camera_pose=pose)["render"].cpu()
            was = ext.render_only(False)
            try:
                i2 = render(cam, g, st.pipe, st.background, camera_pose=pose)["render"].cpu()
            finally:
                ext.render_only(was)
        i3 = render(cam, g, st.pipe, st.background, camera_pose=pose)["render"].detach().cpu()
        assert torch.equal(i1, img_t) and torch.equal(i2, img_t) and torch.equal(i3, img_t)
    # a training render right after a render-only one of the same view still gives the gradients of the training path
    for t in (g._xyz, g._opacity, g.P):
        t.grad = None
    with torch.no_grad():
        render(cam, g, st.pipe, st.background, camera_pose=pose)
    pkg = render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(cam.uid))
    (pkg["render"] * st.gt_images[cam.uid]).sum().backward()
    assert float(g._xyz.grad.abs().max()) > 0 and float(g.P.grad.abs().max()) > 0
    # the operator (all-in-camera-frame inputs): no_grad forward == the forward of a differentiated call
    res = run_blob_case(dev, 600, 80, 48, 2)["dut"]
    with torch.no_grad():
        res_ng = run_blob_case(dev, 600, 80, 48, 2, backward=False)["dut"]
    assert torch.equal(res["color"], res_ng["color"]) and torch.equal(res["radii"], res_ng["radii"])


def check_lazy_loss_expression(dev, H=40, W=56):
    """instantsplat_amd/lazy_loss.py: the reference's loss expression as written (train.py:171-176) served by the loss pair and a
    recorded scalar program.  Against the same expression with the mechanism off (two independent nodes + eager scalar kernels):
    the VALUE must be the bits float32 arithmetic gives for the recorded operations on the pair's two means, the gradient equal
    to rounding; every way out of the recorded form (item, backward twice, mixing with real tensors, long programs, in-place
    edits between the two calls, no_grad) must give what eager PyTorch gives."""
    import numpy as np
    import pytest
    from instantsplat_amd import lazy_loss
    from instantsplat_amd.fused_ssim import fused_ssim
    from instantsplat_amd.loss_utils import l1_loss
    Lz = lazy_loss.LazyScalar
    gen = torch.Generator().manual_seed(4)
    x0 = torch.rand(3, H, W, generator=gen).to(dev)
    gt = (x0 + 0.1 * torch.randn(3, H, W, generator=gen).to(dev)).clamp(0, 1).contiguous()
    f32 = np.float32
    cuda = torch.device(dev).type == "cuda"

    def image():
        leaf = x0.clone().requires_grad_(True)
        return leaf, leaf * 1.0   # a non-leaf like a rendered image

    def eager(expr):
        was, lazy_loss.ENABLED = lazy_loss.ENABLED, False
        try:
            leaf, img = image()
            a, b = l1_loss(img, gt), fused_ssim(img.unsqueeze(0), gt.unsqueeze(0))
            assert type(a) is torch.Tensor and type(b) is torch.Tensor
            v = expr(a, b)
            v.backward()
            return float(v.detach()), leaf.grad.detach().cpu().clone()
        finally:
            lazy_loss.ENABLED = was

    lam = 0.2
    exprs = {
        "train_py": (lambda a, b: (1.0 - lam) * a + lam * (1.0 - b), lambda a, b: f32(f32(1.0 - lam) * a) + f32(f32(lam) * f32(f32(1.0) - b))),
        "l1_only": (lambda a, b: a, lambda a, b: a),
        "ssim_only": (lambda a, b: 1 - b, lambda a, b: f32(1.0) - b),
        "mixed": (lambda a, b: -((a + b) / 2 - 0.1) + (b - a) * 3, lambda a, b: f32(-f32(f32(f32(a + b) * f32(f32(1.0) / f32(2.0))) + f32(-0.1))) + f32(f32(b - a) * f32(3))),
        "twice_l1": (lambda a, b: a + a + 0.5 * b, lambda a, b: f32(f32(a + a) + f32(f32(0.5) * b))),
    }
    for name, (expr, np_expr) in exprs.items():
        leaf, img = image()
        a = l1_loss(img, gt)
        b = fused_ssim(img.unsqueeze(0), gt.unsqueeze(0))
        assert type(a) is Lz and type(b) is Lz, name
        rec = a._rec
        v = expr(a, b)
        assert type(v) is Lz and v._real is None, name           # nothing has been launched for the arithmetic
        v.backward()
        val = v.item()                                            # (after backward, as train.py:188: the materialised tensor is reused)
        assert v._real is not None and isinstance(val, float)
        l1v, ssv = f32(float(rec.l1)), f32(float(rec.ssim))
        assert val == float(np_expr(l1v, ssv)), (name, val, float(np_expr(l1v, ssv)))   # one float32 rounding per recorded operation
        ref_val, ref_grad = eager(expr)
        bound("lazy_loss/value[%s]" % name, abs(val - ref_val), 3e-7)
        bound("lazy_loss/grad[%s]" % name, rel_l2(leaf.grad, ref_grad), 2e-6)
    # ---- the ways out of the recorded form
    leaf, img = image()
    a, b = l1_loss(img, gt), fused_ssim(img.unsqueeze(0), gt.unsqueeze(0))
    loss = 0.8 * a + 0.2 * (1.0 - b)
    with torch.no_grad():
        v0 = loss.item()                      # read before backward, under no_grad: must not spoil the later backward
    assert "tensor(" in repr(loss) and loss.dim() == 0 and loss.dtype == torch.float32 and loss.device.type == torch.device(dev).type
    loss.backward(retain_graph=True)
    g1 = leaf.grad.clone()
    loss.backward()
    assert torch.allclose(leaf.grad, 2 * g1) and abs(loss.item() - v0) == 0.0
    real = torch.tensor(2.0, device=dev)
    leaf, img = image()
    a, b = l1_loss(img, gt), fused_ssim(img.unsqueeze(0), gt.unsqueeze(0))
    mix = a * real + torch.stack([a, b]).sum() + (a < 100)   # a real tensor operand / other functions: ordinary tensors from there on
    assert type(mix) is torch.Tensor
    mix.backward()
    assert float(leaf.grad.abs().sum()) > 0
    long = a
    for _ in range(20):
        long = long * 1.01 + 0.001
    assert type(long) is torch.Tensor      # longer than a program may be (16 operations): ordinary tensors from there on
    chk = f32(float(a._rec.l1))
    for _ in range(20):
        chk = f32(f32(chk * f32(1.01)) + f32(0.001))
    assert float(long.detach()) == float(chk)
    g = torch.autograd.grad(0.5 * a + b, img, retain_graph=True)[0]
    assert g.shape == img.shape and float(g.abs().sum()) > 0
    with pytest.raises(RuntimeError):     # the one entry point that does not consult __torch_function__: a loud failure, not a wrong number
        torch.autograd.backward(0.5 * a)
    # ---- loss.item() (train.py:188) from the pinned slot the program kernel also stored the value in: the tensor's bits, without
    # a copy; a slot that later materialisations have taken over falls back to the tensor; EARLY_ITEM = False never takes one
    from instantsplat_amd import _lib as _l
    leaf, img = image()
    a, b = l1_loss(img, gt), fused_ssim(img.unsqueeze(0), gt.unsqueeze(0))
    first = 0.8 * a + 0.2 * (1.0 - b)
    first.backward()
    assert (first._slot is not None) == (_l.compiled() is not None)
    v_first = first.item()
    assert v_first == float(first._real.detach()) and isinstance(v_first, float)
    for _ in range(lazy_loss._N_SLOTS + 3):
        leaf2, img2 = image()
        a2 = l1_loss(img2, gt)
        (a2 * 2.0).item()
    assert first.item() == v_first and first._real.item() == v_first
    was_early, lazy_loss.EARLY_ITEM = lazy_loss.EARLY_ITEM, False
    try:
        leaf2, img2 = image()
        a2, b2 = l1_loss(img2, gt), fused_ssim(img2.unsqueeze(0), gt.unsqueeze(0))
        late = 0.8 * a2 + 0.2 * (1.0 - b2)
        late.backward()
        assert late._slot is None and late.item() == v_first
    finally:
        lazy_loss.EARLY_ITEM = was_early
    # ---- when the second call is NOT the other half
    leaf, img = image()
    a = l1_loss(img, gt)
    other = fused_ssim(img.unsqueeze(0), (gt * 0.5).unsqueeze(0))
    assert type(other) is torch.Tensor          # another gt
    img2 = img.clone()
    a = l1_loss(img2, gt)
    with torch.no_grad():
        img2.mul_(0.5)                           # the image changed in place between the two calls: its version counter says so
    changed = fused_ssim(img2.unsqueeze(0), gt.unsqueeze(0))
    assert type(changed) is torch.Tensor
    was, lazy_loss.ENABLED = lazy_loss.ENABLED, False
    ssim_half = float(fused_ssim(img2.detach().unsqueeze(0), gt.unsqueeze(0)))
    lazy_loss.ENABLED = was
    bound("lazy_loss/ssim_after_inplace_edit", abs(float(changed) - ssim_half), 1e-6 if cuda else 0.0)
    with torch.no_grad():                        # evaluation (train.py:277-284): no pair, no lazies
        assert type(l1_loss(img, gt)) is torch.Tensor
    assert type(l1_loss(img.detach(), gt)) is torch.Tensor
    assert type(fused_ssim(img.unsqueeze(0), gt.unsqueeze(0), train=False)) is torch.Tensor
    lazy_loss.forget()


def check_deterministic_backward(dev, iters=12, Wm=20, W=64, H=48, min_units=None):
    """mi355gs_tune_deterministic(1): every Gaussian's moments are summed in the order of its tile rectangle instead of with float
    atomics.  (1) The gradients of one frame equal the default path's to rounding, through the operator and through render();
    (2) two backward passes of the same frame are BIT-identical (on the GPU the default path's are not); (3) two training runs
    of the same scene — the drop-in loop and the one-call loop — are bit-identical after `iters` iterations."""
    import instantsplat_amd.diff_gaussian_rasterization as dgr
    from instantsplat_amd import _lib
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import release_trainer, setup_training, train_iteration
    from tests.util import relerr, run_blob_case
    L = _lib.lib()
    old_units = L.mi355gs_tune_min_units(min_units) if min_units else None
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P")
    try:
        base = run_blob_case(dev, 900, 96, 64, 2)["dut"]
        was = dgr.set_deterministic(True)
        assert was is False
        try:
            a = run_blob_case(dev, 900, 96, 64, 2)["dut"]
            b = run_blob_case(dev, 900, 96, 64, 2)["dut"]
            assert torch.equal(a["color"], base["color"]) and torch.equal(a["radii"], base["radii"])
            for k in base["grads"]:
                assert torch.equal(a["grads"][k], b["grads"][k]), k                    # run to run: the same bits
                bound("deterministic/grad_" + k, relerr(a["grads"][k], base["grads"][k]), 2e-6)   # against the atomics: rounding

            def run(fused_step):
                st = generic_start(setup_training(syn_pointmap(3, Wm, Wm, W, H, seed=2), dev))
                losses = [train_iteration(st, fused_step=fused_step) for _ in range(iters)]
                release_trainer(st)
                dgr.BinningPolicy.reset("exact")
                return losses, [getattr(st.gaussians, n).detach().cpu().clone() for n in names]
            for fused_step in (False, True):
                l1, p1 = run(fused_step)
                l2, p2 = run(fused_step)
                assert l1 == l2, (fused_step, l1, l2)
                for n, x, y in zip(names, p1, p2):
                    assert torch.equal(x, y), (fused_step, n)
        finally:
            dgr.set_deterministic(False)
        # ... and the default path is back, untouched
        c = run_blob_case(dev, 900, 96, 64, 2)["dut"]
        for k in base["grads"]:
            bound("deterministic/default_path_after/" + k, relerr(c["grads"][k], base["grads"][k]), 2e-6 if torch.device(dev).type == "cuda" else 0.0)
    finally:
        if old_units:
            L.mi355gs_tune_min_units(old_units)


def check_lazy_scalar_behaves_like_a_tensor(dev, H=40, W=56):
    """What an unmodified caller may do with the values of train.py:171-176 — `Ll1`, `ssim_value`, `loss` — when they are recorded
    expressions (instantsplat_amd/lazy_loss.py) instead of eager tensors: ~50 uses (read-backs and formatting, predicates,
    every backward form, autograd.grad, further arithmetic with numbers / 0-dim / full tensors, in-place ops, stacking, a chain
    longer than the recorded program, a second image pair, copies and pickles, gradient accumulation over two losses), each
    compared with the same lines run with the mechanism off."""
    import copy, math, pickle
    from instantsplat_amd import lazy_loss, loss_utils
    from instantsplat_amd.fused_ssim import fused_ssim
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(0)).to(dev)

    def grad_of(loss, img):
        loss.backward()
        return img.grad.clone()

    def chain(x):
        for _ in range(6):
            x = ((x * 2 + 1) * 0.5) - 0.25
        return x

    uses = {
        "item": lambda l, a, s, i: l.item(), "float": lambda l, a, s, i: float(l), "format": lambda l, a, s, i: f"{l:.5f}",
        "isnan": lambda l, a, s, i: bool(torch.isnan(l)), "math.isfinite": lambda l, a, s, i: math.isfinite(l), "compare": lambda l, a, s, i: bool(l > 0.01),
        "shape": lambda l, a, s, i: (tuple(l.shape), l.dtype, l.device.type, l.dim(), l.numel(), l.requires_grad, l.is_leaf, l.grad_fn is None),
        "backward": lambda l, a, s, i: grad_of(l, i),
        "backward_retain_twice": lambda l, a, s, i: (l.backward(retain_graph=True), l.backward(), i.grad.clone())[2],
        "backward_gradient": lambda l, a, s, i: (l.backward(torch.tensor(2.0, device=i.device)), i.grad.clone())[1],
        "autograd.grad": lambda l, a, s, i: torch.autograd.grad(l, i)[0],
        "item_then_backward": lambda l, a, s, i: (l.item(), grad_of(l, i))[1], "backward_then_item": lambda l, a, s, i: (grad_of(l, i), l.item())[1],
        "Ll1_item_after_backward": lambda l, a, s, i: (grad_of(l, i), a.item())[1], "ssim_item": lambda l, a, s, i: s.item(),
        "detach_cpu": lambda l, a, s, i: l.detach().cpu(), "clone": lambda l, a, s, i: l.clone(), "mean": lambda l, a, s, i: l.mean(),
        "pow": lambda l, a, s, i: l ** 2, "abs": lambda l, a, s, i: abs(l), "neg": lambda l, a, s, i: -l, "div": lambda l, a, s, i: l / 2,
        "rdiv": lambda l, a, s, i: 2 / l, "add_0dim": lambda l, a, s, i: l + torch.tensor(0.5, device=i.device),
        "mul_0dim": lambda l, a, s, i: l * torch.tensor(3.0, device=i.device), "add_image": lambda l, a, s, i: (l + i).sum(),
        "stack": lambda l, a, s, i: torch.stack([l, a]), "iadd": lambda l, a, s, i: l.__iadd__(1.0), "sqrt": lambda l, a, s, i: torch.sqrt(l),
        "psnr_like": lambda l, a, s, i: 20 * torch.log10(1.0 / torch.sqrt(a)), "numpy": lambda l, a, s, i: float(l.detach().cpu().numpy()),
        "tolist": lambda l, a, s, i: l.tolist(), "bool": lambda l, a, s, i: bool(l), "int": lambda l, a, s, i: int(l * 100),
        "repr": lambda l, a, s, i: repr(l).startswith("tensor("), "long_chain": lambda l, a, s, i: grad_of(chain(l), i),
        "second_pair": lambda l, a, s, i: float(l + loss_utils.l1_loss(i * 0.5, gt)), "isinstance": lambda l, a, s, i: isinstance(l, torch.Tensor),
        "deepcopy": lambda l, a, s, i: float(copy.deepcopy(l.detach())), "pickle": lambda l, a, s, i: float(pickle.loads(pickle.dumps(l.detach().cpu()))),
        "to_f64": lambda l, a, s, i: l.to(torch.float64).dtype,
        "two_losses_accumulate": lambda l, a, s, i: (l.backward(), loss_utils.l1_loss(i, gt).backward(), i.grad.clone())[2],
    }

    def same(a, b):
        if isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor):
            return a.shape == b.shape and torch.allclose(a.detach().double().cpu(), b.detach().double().cpu(), rtol=3e-6, atol=1e-7)
        if isinstance(a, float) and isinstance(b, float):
            return abs(a - b) <= 3e-6 * max(1.0, abs(b))
        return a == b

    was = lazy_loss.ENABLED
    try:
        for name, use in uses.items():
            res = []
            for enabled in (True, False):
                lazy_loss.ENABLED = enabled
                img = (gt + 0.1 * torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)).clamp(0, 1).requires_grad_(True)
                Ll1 = loss_utils.l1_loss(img, gt)
                ss = fused_ssim(img.unsqueeze(0), gt.unsqueeze(0))
                loss = (1.0 - 0.2) * Ll1 + 0.2 * (1.0 - ss)
                if enabled:
                    assert type(loss).__name__ == "LazyScalar", type(loss)
                res.append(use(loss, Ll1, ss, img))
            assert same(res[0], res[1]), (name, res[0], res[1])
    finally:
        lazy_loss.ENABLED = was


def check_late_item_of_an_old_loss(dev, H=24, W=32, kept=70):
    """A caller that KEEPS its losses and reads them late (`losses.append(loss)` ... `[l.item() for l in losses]`): the pinned
    word an old loss was promised has been handed to a later one after 64 materialisations (lazy_loss._host_slot).  Its
    `.item()` must notice at once — not spin for its timeout — and return the loss's own value through the ordinary read."""
    import time
    from instantsplat_amd import lazy_loss, loss_utils
    from instantsplat_amd.fused_ssim import fused_ssim
    if not (lazy_loss.ENABLED and lazy_loss.EARLY_ITEM):
        return
    g = torch.Generator().manual_seed(3)
    gt = torch.rand(3, H, W, generator=g).to(dev)
    losses, want = [], []
    for k in range(kept):
        img = (gt + 0.05 * (k + 1) / kept * torch.randn(3, H, W, generator=g).to(dev)).clamp(0, 1).requires_grad_(True)
        Ll1 = loss_utils.l1_loss(img, gt)
        loss = (1.0 - 0.2) * Ll1 + 0.2 * (1.0 - fused_ssim(img.unsqueeze(0), gt.unsqueeze(0)))
        loss.backward()
        losses.append(loss)
        with torch.no_grad():
            d = img.detach()
            want.append(0.8 * float((d - gt).abs().mean()) + 0.2 * (1.0 - float(ssim_ref.ssim(d.unsqueeze(0).cpu(), gt.unsqueeze(0).cpu()))))
    t0 = time.perf_counter()
    got = [l.item() for l in losses]
    dt = time.perf_counter() - t0
    for k in range(kept):
        assert abs(got[k] - want[k]) <= 2e-6, (k, got[k], want[k])
    assert dt < 0.15 * kept / 70 + 0.1, dt   # (six taken-over slots at the old 200 ms spin each would be 1.2 s)
