"""The BASELINE.json configs that are not the bench line, measured and recorded: run on the GPU box,
output -> profiles/r<NN>_baseline_configs.txt.

  configs[0] -> C1'  3-view 128x128-pointmap scene (49,152 Gaussians, 256x256 images), 50 train iterations on the CPU path
                     (oracle/train_ref.py: the plumbing run; MASt3R init is impossible offline) — and the same 50
                     iterations on the device from the same start, loss by loss
  configs[0] -> C1   the same plumbing run on the reference's own example frames (assets/sora/Art, 1280x720 JPEG) through the init-directory
                     loader, device vs CPU oracle loss by loss (tests/sora_util.py: what replaces MASt3R)
  configs[1] -> C2   50k random Gaussians, one 512x512 camera, forward raster only: device ms/frame vs the CPU port, max |d|
  configs[2] -> C3   the bench line (bench.py)
  configs[3] -> C4   12-view pointmap, 995,328 Gaussians, 1920x1080: render + fused loss + backward per view
  configs[4] -> C5   bench.py --gpus 8 (the driver's scaling run; on this 1-GPU box: bench.py --gpus 2 sharing the device)
Measurement helper, not product code."""
import math, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantsplat_amd.synthetic import syn_blob, syn_pointmap
from instantsplat_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, keep_last_frame, last_frame_stats
from instantsplat_amd.train import setup_training, train_iteration
from tests.util import settings_for
from oracle import gs_ref, raster_torch as rt
from oracle.train_ref import CpuTrainer
dev = torch.device('cuda:0')
threads = int(gs_ref.lib().gsref_set_threads(min(os.cpu_count() or 1, 32)))
torch.set_num_threads(threads)
print(f"host: {os.cpu_count()} CPUs visible, CPU port on {threads} threads; device: {torch.cuda.get_device_properties(0).name}")


def timeit(f, n=20, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n


# ---- C1': the plumbing run — 50 iterations on the CPU path, and the same iterations on the device
st = setup_training(syn_pointmap(3, 128, 128, 256, 256, seed=0), dev)
g = st.gaussians
g.update_learning_rate(1)
lrs = {grp["name"]: grp["lr"] for grp in g.optimizer.param_groups}
params = dict(xyz=g._xyz, f_dc=g._features_dc, f_rest=g._features_rest, opacity=g._opacity, scaling=g._scaling, rotation=g._rotation, pose=g.P)
cpu = CpuTrainer(params, st.cameras, st.gt_images, g.per_point_lr, lrs)
l_cpu, l_dev = [], []
t_cpu = t_dev = 0.0
for it in range(50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    l_dev.append(train_iteration(st))                    # reference-shaped loop on the drop-in operators
    torch.cuda.synchronize(); t_dev += time.perf_counter() - t0
    for grp, dgrp in zip(cpu.opt.param_groups, g.optimizer.param_groups):
        grp["lr"] = dgrp["lr"]
    t0 = time.perf_counter()
    l_cpu.append(cpu.iteration())
    t_cpu += time.perf_counter() - t0
worst = max(abs(a - b) / max(abs(b), 1e-2) for a, b in zip(l_dev, l_cpu))
print(f"C1' 3 views / {g.get_xyz.shape[0]} Gaussians / 256^2, 50 train iterations: CPU path {t_cpu:.2f} s ({50 / t_cpu:.2f} it/s), "
      f"device (drop-in loop) {t_dev * 1e3:.1f} ms ({50 / t_dev:.0f} it/s); loss {l_cpu[0]:.5f} -> {l_cpu[-1]:.5f} (CPU), "
      f"{l_dev[0]:.5f} -> {l_dev[-1]:.5f} (device); largest relative loss difference over the 50 iterations {worst:.2e}")
del st, g, cpu

# ---- C1 on the reference's own example frames (assets/sora/Art/images, re-encoded at native size as tests/golden/sora_art): 1280 x 720 JPEGs through the
# init-directory loader; MASt3R (unobtainable offline) is replaced by a synthetic pointmap coloured from the frames + arc poses
import tempfile
from tests import sora_util
from instantsplat_amd import scene_io
from instantsplat_amd.train import evaluate_psnr
with tempfile.TemporaryDirectory() as td:
    sora_util.write_sora_init_dir(os.path.join(td, "Art"), Wm=160, Hm=90)
    sc_ = scene_io.load_init_scene(os.path.join(td, "Art"), 3, resolution=1, device=dev)
    t0 = time.perf_counter()
    l_dev, l_cpu, st_ = sora_util.train_against_cpu_oracle(sc_, dev, iters=50)
    t_both = time.perf_counter() - t0
    worst = max(abs(a - b) / max(abs(b), 1e-2) for a, b in zip(l_dev, l_cpu))
    from instantsplat_amd.train import train_iteration as _ti
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): _ti(st_, fused_loss=False)
    torch.cuda.synchronize(); t_dev = time.perf_counter() - t0
    print(f"C1 sora/Art: 3 JPEG frames 1280x720 (native, -r 1), {st_.gaussians.get_xyz.shape[0]} Gaussians (synthetic pointmap 160x90 per view, coloured "
          f"from the frames; MASt3R replaced), 50 train iterations, train.py loss as written: loss {l_dev[0]:.5f} -> {l_dev[-1]:.5f} (device), "
          f"{l_cpu[0]:.5f} -> {l_cpu[-1]:.5f} (CPU oracle), largest relative loss difference {worst:.2e}; device loop alone, next 50 iterations: "
          f"{t_dev * 1e3:.1f} ms ({50 / t_dev:.0f} it/s); PSNR {evaluate_psnr(st_):.2f} dB")
    del st_, sc_

# ---- C2: 50k blob, 512^2, forward only, GPU vs CPU port
sc = syn_blob(50000, 512, 512, seed=0)
stg = settings_for(sc.camera, 3, GaussianRasterizationSettings, sc.bg, device=dev)
args = dict(means3D=sc.means3D.to(dev), means2D=torch.zeros(50000, 3, device=dev), opacities=torch.sigmoid(sc.opacity_logit).to(dev),
            shs=sc.shs.to(dev), scales=torch.exp(sc.scaling_logit).to(dev), rotations=sc.rotation.to(dev))
r = GaussianRasterizer(stg)
with torch.no_grad():
    gpu_ms = timeit(lambda: r(**args), n=200)
    keep_last_frame(True)
    img_gpu = r(**args)[0].cpu()
    R, Reff = last_frame_stats(); keep_last_frame(False)
stc = settings_for(sc.camera, 3, rt.RasterSettings, sc.bg)
for _ in range(2):
    img_cpu, _, ctx = gs_ref.forward(sc.means3D, torch.sigmoid(sc.opacity_logit).reshape(-1), stc, shs=sc.shs, scales=torch.exp(sc.scaling_logit), rotations=sc.rotation)
t0 = time.perf_counter()
for _ in range(10):
    img_cpu, _, ctx = gs_ref.forward(sc.means3D, torch.sigmoid(sc.opacity_logit).reshape(-1), stc, shs=sc.shs, scales=torch.exp(sc.scaling_logit), rotations=sc.rotation)
cpu_ms = 1e3 * (time.perf_counter() - t0) / 10
d = (img_gpu - img_cpu).abs()
print(f"C2 50k Gaussians / 512^2 / SH degree 3, forward raster only: device {gpu_ms:.3f} ms/frame (operator call incl. its blocking count "
      f"read-back), CPU port ({threads} threads) {cpu_ms:.1f} ms/frame -> {cpu_ms / gpu_ms:.0f}x; R = {R} (oracle {ctx.num_rendered}), "
      f"R_eff = {Reff}; image max |d| {float(d.max()):.2e}, values off by > 1e-4: {float((d > 1e-4).float().mean()):.2e}")

# ---- C4: 12-view pointmap 288^2 -> 995k Gaussians, 1920x1080, render + loss + backward
from instantsplat_amd.gaussian_renderer import render
from instantsplat_amd.fused_ssim import fused_l1_ssim_loss
scene = syn_pointmap(12, 288, 288, 1920, 1080, seed=0)
t0 = time.perf_counter(); stt = setup_training(scene, dev); torch.cuda.synchronize()
print(f"C4 setup (kNN x2 + 12 teacher renders) {time.perf_counter() - t0:.2f} s, P = {stt.gaussians.get_xyz.shape[0]}")
g = stt.gaussians; cam = stt.cameras[5]


def fb():
    img = render(cam, g, stt.pipe, stt.background, camera_pose=g.get_RT(cam.uid))["render"]
    loss, _ = fused_l1_ssim_loss(img.unsqueeze(0), stt.gt_images[cam.uid].unsqueeze(0), 0.2)
    loss.backward()
    for p in (g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation, g.P): p.grad = None


ms = timeit(fb, n=20)
keep_last_frame(True)
with torch.no_grad(): render(cam, g, stt.pipe, stt.background, camera_pose=g.get_RT(cam.uid))
R, Reff = last_frame_stats(); keep_last_frame(False)
t_full = timeit(lambda: train_iteration(stt), n=24)
print(f"C4 995,328 Gaussians / 1920x1080, one view: render + fused loss + backward {ms:.3f} ms; full train iteration (drop-in loop, "
      f"12 views round-robin) {t_full:.3f} ms = {1e3 / t_full:.0f} it/s; R = {R}, R_eff = {Reff}")
