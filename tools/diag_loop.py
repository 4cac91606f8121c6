"""Hardware diagnostic for the reference-trajectory tests (tests/test_zz_reference_functions_gpu.py).

Teacher forcing: at each iteration the parameters of the reference's own `training()` run (tests/golden, `loop_iter_params_*`)
are loaded, one forward + backward runs on the device, and every tensor's gradient is compared with the gradient the
reference run produced at that iteration (`loop_iter_grads_*`).  Then the free-running loop is compared with the recorded
parameters after every iteration, which shows at which iteration and in which tensor a trajectory leaves the golden one.

    python tools/diag_loop.py [cuda|cpu-emu]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import ops_util  # noqa: E402

NAMES = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P")


def key(prefix, kind, n):
    return prefix + kind + (n if n.startswith("_") else "_" + n)


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def teacher_forced(dev, run="loop", fused_loss=True, verbose=True):
    from instantsplat_amd.fused_ssim import fused_l1_ssim_loss, fused_ssim
    from instantsplat_amd.gaussian_renderer import render
    G, st, _ = ops_util._reference_loop_start(dev, run)
    g = st.gaussians
    iters = int(G[run + "_flags"][2])
    worst = {}
    for it in range(iters):
        with torch.no_grad():
            for n in NAMES:
                getattr(g, n).copy_(torch.from_numpy(G[key(run, "_iter_params", n)][it]).to(dev))
                getattr(g, n).grad = None
        uid = int(G[run + "_view_uids"][it])
        cam = st.cameras[uid]
        pkg = render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(uid))
        image, gt = pkg["render"], st.gt_images[uid]
        if fused_loss:
            loss, _ = fused_l1_ssim_loss(image.unsqueeze(0), gt.unsqueeze(0), st.opt.lambda_dssim)
        else:
            loss = 0.8 * (image - gt).abs().mean() + 0.2 * (1.0 - fused_ssim(image.unsqueeze(0), gt.unsqueeze(0)))
        loss.backward()
        row = []
        for n in NAMES:
            ref = torch.from_numpy(G[key(run, "_iter_grads", n)][it])
            got = getattr(g, n).grad
            got = torch.zeros_like(ref) if got is None else got.detach().cpu()
            r = rel(got, ref) if float(ref.abs().max()) > 0 else float(got.abs().max())
            worst[n] = max(worst.get(n, 0.0), r)
            row.append("%s %.2e" % (n, r))
        if verbose:
            print("  it %2d view %d loss %.7f (golden %.7f, rel %.1e)  grad rel-L2: %s" % (
                it, uid, float(loss), G[run + "_losses"][it], abs(float(loss) - G[run + "_losses"][it]) / G[run + "_losses"][it],
                "  ".join(row)))
    return worst


def free_running(dev, fused_step, run="loop"):
    from instantsplat_amd.train import train_iteration
    G, st, _ = ops_util._reference_loop_start(dev, run)
    g = st.gaussians
    iters = int(G[run + "_flags"][2])
    for it in range(iters):
        l = float(train_iteration(st, fused_step=fused_step))
        row = []
        for n in NAMES:
            ref = G[key(run, "_iter_params", n)][it + 1] if it + 1 < iters else G[key(run, "_final", n)]
            row.append("%s %.2e" % (n, rel(getattr(g, n).detach().cpu(), torch.from_numpy(ref))))
        print("  it %2d loss %.7f (golden %.7f)  param rel-L2 after step: %s" % (it, l, G[run + "_losses"][it], "  ".join(row)))
    # element-level view of the worst tensor
    for n in NAMES:
        a, b = getattr(g, n).detach().cpu().double().flatten(), torch.from_numpy(G[key(run, "_final", n)]).double().flatten()
        d = (a - b).abs()
        print("  final %-15s rel %.2e  max|d| %.2e  #>1e-4: %d / %d  median|d| %.2e" % (
            n, rel(a, b), float(d.max()), int((d > 1e-4).sum()), d.numel(), float(d.median())))


if __name__ == "__main__":
    dev = sys.argv[1] if len(sys.argv) > 1 else "cuda"
    if dev == "cpu-emu":
        from instantsplat_amd import _lib
        import subprocess
        subprocess.check_call(["bash", os.path.join(ROOT, "tests", "emu", "build_emu.sh")])
        _lib._use_library_for_testing(os.path.join(ROOT, "tests", "emu", "libmi355gs_emu.so"))
        dev = "cpu"
    for run in ("loop", "loopb"):
        for fl in (True, False):
            print("== teacher-forced, run=%s, fused_loss=%s" % (run, fl))
            w = teacher_forced(dev, run, fl)
            print("   worst per tensor:", {k: "%.2e" % v for k, v in w.items()})
    for run in ("loop", "loopb"):
        for fs in ((False, True) if run == "loop" else (False,)):
            print("== free-running, run=%s, fused_step=%s" % (run, fs))
            free_running(dev, fs, run)
