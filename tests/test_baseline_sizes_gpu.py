"""GPU tier, oracle parity AT the BASELINE.json sizes (VERDICT r1 "missing" #3): the C oracle (oracle/gs_ref.c, OpenMP) does
200k Gaussians at 512x512 in a fraction of a second and 1 M at 1080p in seconds, so the full-size configurations are
compared with it directly — forward image, radii and every gradient — not only through size-independent properties
(tests/test_properties_gpu.py keeps those).

  blob-200k   SYN-BLOB(200000, 512, 512)            rasterizer operator alone, SH degree 0 and 3          
  C3          SYN-POINTMAP(3, 256, 256, 512, 512)   196,608 Gaussians, all 3 views, through render() + L1/SSIM loss:
              the student at iteration 0 (isotropic create_from_pcd state) and after 40 training iterations on the device
  C4          SYN-POINTMAP(12, 288, 288, 1920, 1080) 995,328 Gaussians, 1080p, one view, same comparison

What "equal" means at this size (measured on MI355X, round 2; tools/diag_fullsize.py and GS_CALIBRATE=1 on this file).
Three implementations are run on the same inputs — device (fp32), C oracle fp32, C oracle fp64 — because at 10^8 (pixel,
Gaussian) pair evaluations per frame NO fp32 implementation meets the small-case tolerance (1e-4) against fp64:
  * the algorithm is discontinuous where a pair is skipped (alpha < 1/255) and where a pixel stops (T < 1e-4); a handful of
    pairs land within float rounding of a threshold, and each flip moves ONE Gaussian's gradient by a few per cent of itself
    (blob-200k, linear loss: device vs fp32 oracle has 100 % of the squared gradient error in its 10 worst Gaussians, the
    fp32 oracle vs fp64 92 %; with the 64 worst of 200,000 set aside every gradient agrees to 5e-6);
  * the training loss adds sign(image - gt) (L1): pixels the model already matches to float rounding flip sign between ANY
    two implementations, which moves every Gaussian under them.  That is why the errors grow after 40 training iterations
    (C3 view 0: xyz 2.6e-3 for the device AND 2.6e-3 for the fp32 oracle, both against fp64).
So the assertions are relative to what the fp32 oracle itself achieves against fp64:
  * image: every value within 5e-3 of fp64; fraction of values off by more than 1e-4 at most FACTOR x the fp32 oracle's (floor 2e-4);
  * per-Gaussian gradient tensors: relative L2 against fp64 at most FACTOR x the fp32 oracle's (floor 1e-3), also with the 64
    worst Gaussians set aside (floor 1e-4).  Round 5: the projection kernels are compiled without FMA contraction and round like
    the oracle (instantsplat_amd/csrc/Makefile), which took most of the spread out of these ratios — the largest of this
    file's 80 ratios fell to 1.62 and the image ratios to 1.00-1.12 — so FACTOR = 2.5 (was 4) for the cases with fixed
    inputs (blob-200k, C3 / C4 at iteration 0, and every case in the deterministic-backward mode, whose trained state repeats
    bit for bit); for the trained states of the default mode (which pairs flip is luck there: the trained state itself differs
    from run to run) 3 on whole tensors (was 8) and 4 on the outlier-free statistics (was 5), from 14 repeated runs committed
    as profiles/r05_trained_state_ratio_distribution.txt (round 4's 30 runs: r04_...); image fractions 2 everywhere.  Every
    measured ratio is printed (pytest -s) and recorded (GS_CALIBRATE=1).  A wrong kernel is off by orders of magnitude;
  * pose gradients (sums over all Gaussians), loss: plain relative bounds.
"""
import pytest
import torch

from tests import ops_util
from tests.ops_util import bound

pytestmark = pytest.mark.gpu
OUTLIERS = 64
FACTOR_FIXED_INPUTS = 2.5   # device error allowed, in units of the fp32 oracle's own error against fp64 (see the module docstring)
# Trained states, from the distribution over 14 repeated runs of the C3 case and 2 of the C4 case on the round-5 tree
# (profiles/r05_trained_state_ratio_distribution.txt, tools/trained_state_ratios.py; worst tensor and view of each run):
#   whole gradient tensors      median 1.12, 90th percentile 1.22, max 1.42   (round 4: 1.37 / 1.9 / 5.75)  -> 3
#   without the 64 worst rows   median 1.27, 90th percentile 1.44, max 2.18   (round 4: 1.39 / 1.9 / 2.64)  -> 4
#   image values off by > 1e-4  median 1.01, 90th percentile 1.01, max 1.03   (round 4: 1.17 / 1.8 / 2.39)  -> 2
FACTOR_TRAINED_STATE = 3.0
FACTOR_TRAINED_STATE_ROBUST = 4.0   # (~2x the largest of the runs: a red GPU tier on a rare run teaches nothing)
FACTOR_IMAGE = 2.0


def _grad_errors(a, b):
    """(relative L2, relative L2 without the OUTLIERS rows of largest squared error) of a against b; rows = Gaussians."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    e = ((a - b) ** 2).reshape(a.shape[0], -1).sum(1)
    n = float(b.norm()) + 1e-300
    full = float(e.sum().sqrt()) / n
    if e.numel() <= OUTLIERS:
        return full, full
    keep = torch.sort(e).values[:-OUTLIERS]
    return full, float(keep.sum().sqrt()) / n


def _robust(factor):
    """the limit for the outlier-free statistics (image fraction, gradients without the 64 worst Gaussians)"""
    return FACTOR_TRAINED_STATE_ROBUST if factor == FACTOR_TRAINED_STATE else factor


def _check_image(pre, dut, c32, c64, factor):
    factor = min(_robust(factor), FACTOR_IMAGE)
    d = (dut.detach().double().cpu() - c64).abs()
    d_ref = (c32.detach().double() - c64).abs()
    bound(pre + "image_max", float(d.max()), 5e-3)
    frac, frac_ref = float((d > 1e-4).double().mean()), float((d_ref > 1e-4).double().mean())
    print("%-46s device %.2e  fp32 oracle %.2e  ratio %.2f (limit %g)" % (pre + "image_frac_over_1e-4", frac, frac_ref,
                                                                          frac / max(frac_ref, 1e-30), factor))
    bound(pre + "image_frac_over_1e-4[fp32 oracle: %.1e]" % frac_ref, frac, max(factor * frac_ref, 2e-4))


def _check_grad(pre, k, dut, c32, c64, factor):
    full, robust = _grad_errors(dut, c64)
    full_ref, robust_ref = _grad_errors(c32, c64)
    print("%-46s device %.2e  fp32 oracle %.2e  ratio %.2f | without the %d worst: %.2e / %.2e  ratio %.2f (limit %g)" % (
        pre + "grad_" + k, full, full_ref, full / max(full_ref, 1e-30), OUTLIERS, robust, robust_ref, robust / max(robust_ref, 1e-30), factor))
    bound(pre + "grad_%s[fp32 oracle: %.1e]" % (k, full_ref), full, max(factor * full_ref, 1e-3))
    bound(pre + "grad_%s_without_%d_worst[fp32 oracle: %.1e]" % (k, OUTLIERS, robust_ref), robust, max(_robust(factor) * robust_ref, 1e-4))


@pytest.mark.parametrize("deg", [0, 3])
def test_blob_200k_512_forward_backward_matches_oracle(gpu, deg):
    from instantsplat_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from instantsplat_amd.synthetic import syn_blob
    from oracle import gs_ref, raster_torch as rt
    from tests.util import settings_for
    P, W, H = 200000, 512, 512
    gs_ref.lib().gsref_set_threads(32)
    sc = syn_blob(P, W, H, seed=0, scale_mean=0.02)
    torch.manual_seed(100)
    wgt = torch.randn(3, H, W)
    bg = torch.tensor([0.2, 0.5, 0.9])
    res = {}
    for which in ("dut", "c32", "c64"):
        dt = torch.float64 if which == "c64" else torch.float32
        dev = gpu if which == "dut" else torch.device("cpu")
        lv = dict(means3D=sc.means3D, scaling=sc.scaling_logit, rot=sc.rotation, op=sc.opacity_logit, shs=sc.shs)
        lv = {k: v.clone().to(dt).to(dev).requires_grad_(True) for k, v in lv.items()}
        m2d = torch.zeros(P, 3, dtype=dt, device=dev, requires_grad=True)
        kw = dict(shs=lv["shs"], scales=torch.exp(lv["scaling"]), rotations=lv["rot"])
        if which == "dut":
            st = settings_for(sc.camera, deg, GaussianRasterizationSettings, bg, device=dev)
            color, radii = GaussianRasterizer(st)(means3D=lv["means3D"], means2D=m2d, opacities=torch.sigmoid(lv["op"]), **kw)
        else:
            st = settings_for(sc.camera, deg, rt.RasterSettings, bg)
            if which == "c64":
                st = rt.RasterSettings(*[(x.double() if isinstance(x, torch.Tensor) else x) for x in st])
            color, radii = gs_ref.rasterize(lv["means3D"], m2d, torch.sigmoid(lv["op"]), st, **kw)
        (color * wgt.to(dt).to(dev)).sum().backward()
        res[which] = dict(color=color.detach().cpu().double(), radii=radii.cpu(),
                          grads={**{k: v.grad.detach().cpu() for k, v in lv.items()}, "means2D": m2d.grad.detach().cpu()})
    pre = "blob200k/sh%d/" % deg
    assert int((res["c64"]["radii"] > 0).sum()) > 150000
    mism = res["dut"]["radii"] != res["c32"]["radii"]
    bound(pre + "radii_mismatch_frac", float(mism.float().mean()), 1e-4)
    assert int((res["dut"]["radii"] - res["c32"]["radii"]).abs().max()) <= 1
    _check_image(pre, res["dut"]["color"], res["c32"]["color"], res["c64"]["color"], FACTOR_FIXED_INPUTS)
    for k in res["c64"]["grads"]:
        _check_grad(pre, k, res["dut"]["grads"][k], res["c32"]["grads"][k], res["c64"]["grads"][k], FACTOR_FIXED_INPUTS)


def _compare_views_with_cpu_oracle(gpu, tag, V, Wm, W, H, views, train_iters, deterministic=False):
    """deterministic: the run (training and the compared backward) in the deterministic-backward mode (include/mi355gs.h,
    mi355gs_tune_deterministic).  The trained state then is the SAME state every time the test runs — which pairs sit on a
    threshold is no longer luck — so it is held to the fixed-input factor."""
    import instantsplat_amd.diff_gaussian_rasterization as dgr
    if deterministic:
        assert dgr.set_deterministic(True) is False
        try:
            return _compare_views_with_cpu_oracle(gpu, tag + "_det", V, Wm, W, H, views, train_iters)
        finally:
            dgr.set_deterministic(False)
    from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
    from instantsplat_amd.fused_ssim import fused_l1_ssim_loss
    from instantsplat_amd.gaussian_renderer import render
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import RunAhead, setup_training
    from oracle import gs_ref
    from oracle.ssim_ref import l1_loss, ssim
    from oracle.train_ref import CpuTrainer
    gs_ref.lib().gsref_set_threads(32)
    st = setup_training(syn_pointmap(V, Wm, Wm, W, H, seed=0), gpu)
    g = st.gaussians
    try:
        if train_iters:
            ra = RunAhead(st, window=10)
            for _ in range(train_iters):
                ra.step()
            ra.flush()
            torch.cuda.synchronize()
    finally:
        BinningPolicy.reset("exact")
    names = dict(xyz="_xyz", f_dc="_features_dc", f_rest="_features_rest", opacity="_opacity", scaling="_scaling",
                 rotation="_rotation", pose="P")
    params = {k: getattr(g, n) for k, n in names.items()}
    for p in params.values():
        p.grad = None
    lrs = {grp["name"]: grp["lr"] for grp in g.optimizer.param_groups}
    cpus = {}
    for which, dt in (("c32", torch.float32), ("c64", torch.float64)):
        cpus[which] = CpuTrainer({k: v.detach().cpu().to(dt) for k, v in params.items()}, st.cameras,
                                 [x.detach().cpu().to(dt) for x in st.gt_images], g.per_point_lr, lrs, sh_degree=g.active_sh_degree)
    for uid in views:
        cam = st.cameras[uid]
        img = render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(uid))["render"]
        loss, _ = fused_l1_ssim_loss(img.unsqueeze(0), st.gt_images[uid].unsqueeze(0), st.opt.lambda_dssim)
        loss.backward()
        ref = {}
        for which, cpu in cpus.items():
            img_c = cpu.render(cam, cpu.p["pose"][uid])
            gt_c = cpu.gts[uid]
            loss_c = 0.8 * l1_loss(img_c, gt_c) + 0.2 * (1.0 - ssim(img_c.unsqueeze(0), gt_c.unsqueeze(0)))
            loss_c.backward()
            ref[which] = dict(img=img_c.detach().double(), loss=float(loss_c.detach()), grads={k: cpu.p[k].grad.clone() for k in params})
            for k in params:
                cpu.p[k].grad = None
        pre = "%s/it%d/view%d/" % (tag, train_iters, uid)
        factor = FACTOR_TRAINED_STATE if (train_iters and not dgr._lib.lib().mi355gs_tune_deterministic(-1)) else FACTOR_FIXED_INPUTS
        bound(pre + "loss", abs(float(loss.detach()) - ref["c64"]["loss"]) / abs(ref["c64"]["loss"]), 5e-5)
        _check_image(pre, img, ref["c32"]["img"], ref["c64"]["img"], factor)
        g64 = ref["c64"]["grads"]
        for k, t in params.items():
            a = t.grad.detach().cpu()
            if k == "f_rest" and g.active_sh_degree == 0:
                assert float(a.abs().max()) == 0.0 and float(g64[k].abs().max()) == 0.0
            elif k == "pose":
                e = float((a.double() - g64[k]).norm() / g64[k].norm())
                e_ref = float((ref["c32"]["grads"][k].double() - g64[k]).norm() / g64[k].norm())
                bound(pre + "grad_pose[fp32 oracle: %.1e]" % e_ref, e, 1e-3)
            else:
                assert float(g64[k].norm()) > 0, k
                _check_grad(pre, k, a, ref["c32"]["grads"][k], g64[k], factor)
            t.grad = None


@pytest.mark.parametrize("train_iters,deterministic", [(0, False), (40, False), (40, True)])
def test_c3_196k_512_all_views_match_cpu_oracle(gpu, train_iters, deterministic):
    _compare_views_with_cpu_oracle(gpu, "C3", 3, 256, 512, 512, (0, 1, 2), train_iters, deterministic)


@pytest.mark.parametrize("run_ahead", [True, False], ids=["one_call_loop", "dropin_loop"])
def test_c3_deterministic_mode_two_runs_are_bit_identical_after_200_iterations(gpu, run_ahead):
    """SURVEY.md 5 (determinism self-check): in the deterministic-backward mode two runs of C3 (196,608 Gaussians, 512^2, joint pose
    + Gaussian optimisation) hold the same bits after 200 iterations — every parameter, every pose, the loss — on both loops; in
    the default mode (float atomics) two such runs are 0.6 dB apart by then (profiles/r04_rerun_psnr_spread_200_iterations.txt)."""
    import instantsplat_amd.diff_gaussian_rasterization as dgr
    from instantsplat_amd.synthetic import syn_pointmap
    from instantsplat_amd.train import training
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P")
    from instantsplat_amd import lazy_loss
    assert dgr.set_deterministic(True) is False
    try:
        runs = []
        for k in range(3):
            # the third run of the drop-in loop reads `loss.item()` the ordinary way (copy + wait for everything enqueued) instead of
            # from the pinned word the loss kernel writes: the host then never runs ahead of the backward — and nothing may change
            was, lazy_loss.EARLY_ITEM = lazy_loss.EARLY_ITEM, (k < 2)
            try:
                r = training(syn_pointmap(3, 256, 256, 512, 512, seed=0), gpu, iterations=200, run_ahead=run_ahead, fused_loss=run_ahead)
            finally:
                lazy_loss.EARLY_ITEM = was
            runs.append((r["last_loss"], r["psnr_after"], [getattr(r["state"].gaussians, n).detach().clone() for n in names]))
            dgr.BinningPolicy.reset("exact")
    finally:
        dgr.set_deterministic(False)
    print("deterministic mode, 200 iterations of C3 (%s): PSNR %.4f / %.4f dB, last loss %.9f / %.9f" % (
        "one-call loop" if run_ahead else "drop-in loop, train.py loss as written", runs[0][1], runs[1][1], runs[0][0], runs[1][0]))
    for other in runs[1:]:
        assert runs[0][0] == other[0] and runs[0][1] == other[1]
        for n, a, b in zip(names, runs[0][2], other[2]):
            assert torch.equal(a, b), n
    assert runs[0][1] > 30.0


@pytest.mark.parametrize("train_iters", [0, 12])
def test_c4_995k_1080p_matches_cpu_oracle(gpu, train_iters):
    _compare_views_with_cpu_oracle(gpu, "C4", 12, 288, 1920, 1080, (5,), train_iters)


@pytest.mark.parametrize("V,Wm", [(3, 256), (12, 288)], ids=["C3_196608", "C4_995328"])
def test_knn_on_the_baseline_point_clouds(gpu, V, Wm):
    """a12 (create_from_pcd's distCUDA2) at the sizes the configs use: exact 3-NN mean squared distance vs the float64 k-d tree."""
    ops_util.check_knn_pointmap(gpu, V, Wm)
