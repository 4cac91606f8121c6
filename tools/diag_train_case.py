"""A `train` case of tools/fuzz_ops.py outside its bound (seed 41 on the GPU: xyz gradient 1.05e-3 from the fp32 CPU oracle's, bound
1e-4; the same 20 scenes all pass under the emulator): whose error is it?  For every scene of that seed: the first frame's
gradients on the device, from the fp32 CPU oracle and from the SAME oracle in float64 — per tensor, relative L2 of each pair.
python tools/diag_train_case.py [gpu]      Test tooling, not product code."""
import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantsplat_amd import _lib
gpu = len(sys.argv) > 1 and sys.argv[1] == "gpu"
if not gpu:
    _lib._use_library_for_testing(os.path.join(ROOT, "tests", "emu", "libmi355gs_emu.so"))
from instantsplat_amd.fused_ssim import fused_l1_ssim_loss
from instantsplat_amd.gaussian_renderer import render
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import setup_training
dev = torch.device("cuda:0" if gpu else "cpu")
cases = [(2, 27, 70), (27, 19, 48), (31, 16, 70), (38, 24, 24), (52, 26, 16), (54, 19, 70), (58, 13, 33), (65, 23, 16), (71, 20, 96), (76, 9, 24), (86, 8, 48),
         (92, 18, 70), (97, 16, 96), (101, 19, 70), (111, 9, 16), (121, 27, 96), (122, 18, 70), (133, 8, 70), (135, 14, 24), (138, 27, 33)]
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def oracle_grads(params, cam, gt, dt):
    from tests.ops_util import oracle_frame_grads
    return oracle_frame_grads(params, cam, gt, dt), None


for idx, Wm, W in cases:
    st = setup_training(syn_pointmap(3, Wm, Wm, W, W, seed=3), dev)
    g = st.gaussians
    params = dict(xyz=g._xyz, f_dc=g._features_dc, f_rest=g._features_rest, opacity=g._opacity, scaling=g._scaling, rotation=g._rotation, pose=g.P)
    cam = st.cameras[1]
    img = render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(cam.uid))["render"]
    loss, _ = fused_l1_ssim_loss(img.unsqueeze(0), st.gt_images[cam.uid].unsqueeze(0), 0.2)
    loss.backward()
    g32, _ = oracle_grads(params, cam, st.gt_images[cam.uid], torch.float32)
    g64, _ = oracle_grads(params, cam, st.gt_images[cam.uid], torch.float64)
    row = []
    for name, t in params.items():
        if name in ("rotation", "f_rest") or float(g64[name].norm()) == 0:
            continue
        d = t.grad.detach().cpu()
        row.append(f"{name}: dev-f32 {rel(d, g32[name]):.1e} dev-f64 {rel(d, g64[name]):.1e} f32-f64 {rel(g32[name], g64[name]):.1e}")
    gt_c, im = st.gt_images[cam.uid].cpu().double(), img.detach().cpu().double()
    near_zero = int(((im - gt_c).abs() < 2e-7).sum())
    print(f"case {idx} Wm {Wm} W {W} ({3 * Wm * Wm} Gaussians): values with |image - gt| < 2e-7 (sign of the L1 term decided by rounding): {near_zero}; " + "; ".join(row), flush=True)
