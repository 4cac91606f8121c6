"""BASELINE configs[3] (C4): 12 views, 995,328 Gaussians, 1920x1080 — render + fused L1/SSIM loss + backward."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import setup_training
from instantsplat_amd.gaussian_renderer import render
from instantsplat_amd.fused_ssim import fused_l1_ssim_loss
from instantsplat_amd import _lib
if os.environ.get("GS_MIN_UNITS"):   # A/B of the backward's unit length (mi355gs_tune_min_units)
    _lib.lib().mi355gs_tune_min_units(int(os.environ["GS_MIN_UNITS"]))
dev = torch.device('cuda:0')
st = setup_training(syn_pointmap(12, 288, 288, 1920, 1080, seed=0), dev)
g = st.gaussians
def fb(cam):
    img = render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(cam.uid))["render"]
    loss, _ = fused_l1_ssim_loss(img.unsqueeze(0), st.gt_images[cam.uid].unsqueeze(0), 0.2)
    loss.backward()
    for p in (g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation, g.P): p.grad = None
for i in range(3): fb(st.cameras[i])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(12): fb(st.cameras[i])
torch.cuda.synchronize(); print(f"C4 render+loss+backward: {1e3*(time.perf_counter()-t0)/12:.3f} ms / view")
# the render-only forward on the same views (every no-grad render: mi355gs_raster_forward_render_only), and the deterministic backward
with torch.no_grad():
    for i in range(3): render(st.cameras[i], g, st.pipe, st.background, camera_pose=g.get_RT(i))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(12): render(st.cameras[i], g, st.pipe, st.background, camera_pose=g.get_RT(i))
    torch.cuda.synchronize(); print(f"C4 render, no grad (render-only forward): {1e3*(time.perf_counter()-t0)/12:.3f} ms / view")
if os.environ.get("GS_C4_DET", "1") == "1":
    import instantsplat_amd.diff_gaussian_rasterization as dgr
    dgr.set_deterministic(True)
    for i in range(3): fb(st.cameras[i])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(12): fb(st.cameras[i])
    torch.cuda.synchronize(); print(f"C4 render+loss+backward, deterministic-backward mode: {1e3*(time.perf_counter()-t0)/12:.3f} ms / view")
    dgr.set_deterministic(False)
