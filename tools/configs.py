"""Ad-hoc measurement of the other BASELINE configs (not bench lines): C2 forward-only, C4 1M/1080p."""
import math, sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from instantsplat_amd.synthetic import syn_blob, syn_pointmap
from instantsplat_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, keep_last_frame, last_frame_stats
from tests.util import settings_for
from oracle import gs_ref, raster_torch as rt
dev = torch.device('cuda:0')
def timeit(f, n=20, w=3):
    for _ in range(w): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
# ---- C2: 50k blob, 512^2, forward only, GPU vs CPU port
sc = syn_blob(50000, 512, 512, seed=0)
st = settings_for(sc.camera, 3, GaussianRasterizationSettings, sc.bg, device=dev)
args = dict(means3D=sc.means3D.to(dev), means2D=torch.zeros(50000, 3, device=dev), opacities=torch.sigmoid(sc.opacity_logit).to(dev),
            shs=sc.shs.to(dev), scales=torch.exp(sc.scaling_logit).to(dev), rotations=sc.rotation.to(dev))
r = GaussianRasterizer(st)
with torch.no_grad():
    gpu_ms = timeit(lambda: r(**args))
    img_gpu = r(**args)[0].cpu()
stc = settings_for(sc.camera, 3, rt.RasterSettings, sc.bg)
gs_ref.lib().gsref_set_threads(32)
t0 = time.perf_counter()
for _ in range(3):
    img_cpu, _, ctx = gs_ref.forward(sc.means3D, torch.sigmoid(sc.opacity_logit).reshape(-1), stc, shs=sc.shs, scales=torch.exp(sc.scaling_logit), rotations=sc.rotation)
cpu_ms = 1e3 * (time.perf_counter() - t0) / 3
print(f"C2 50k/512^2 SH3 forward: GPU {gpu_ms:.3f} ms/frame, CPU port (32 thr) {cpu_ms:.1f} ms/frame, max|d| {float((img_gpu - img_cpu).abs().max()):.2e}")
# ---- C4: 12-view pointmap 288^2 -> 995k Gaussians, 1920x1080, render + backward
from instantsplat_amd.train import setup_training
from instantsplat_amd.gaussian_renderer import render
from instantsplat_amd.fused_ssim import fused_l1_ssim_loss
scene = syn_pointmap(12, 288, 288, 1920, 1080, seed=0)
t0 = time.perf_counter(); stt = setup_training(scene, dev); torch.cuda.synchronize(); print(f"C4 setup (kNN x2 + 12 teacher renders) {time.perf_counter()-t0:.2f} s, P={stt.gaussians.get_xyz.shape[0]}")
g = stt.gaussians; cam = stt.cameras[5]
def fb():
    img = render(cam, g, stt.pipe, stt.background, camera_pose=g.get_RT(cam.uid))["render"]
    loss, _ = fused_l1_ssim_loss(img.unsqueeze(0), stt.gt_images[cam.uid].unsqueeze(0), 0.2)
    loss.backward()
    for p in (g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation, g.P): p.grad = None
ms = timeit(fb, n=10)
keep_last_frame(True)
with torch.no_grad(): render(cam, g, stt.pipe, stt.background, camera_pose=g.get_RT(cam.uid))
R, Reff = last_frame_stats(); keep_last_frame(False)
print(f"C4 1M/1080p render+loss+backward: {ms:.3f} ms, R={R}, R_eff={Reff}")
