"""A/B of the two instantiations of the forward's stage 2 on FIXED frames: a C3 scene trained for 200 iterations, then N renders of
every training view through the training instantiation (grad mode on, no backward) and N through the render-only one
(torch.no_grad()) — the same sorted lists, the same images (checked here bit for bit).  Prints a JSON line with the kernel
times of both (HIP events on the launch stream, mi355gs_profile_* kinds 0 and 2) and the whole-render wall times; run under
`bash tools/pmc.sh render_only FETCH_SIZE python tools/render_only_loop.py` (and WRITE_SIZE) for the per-instantiation HBM
traffic (k_composite_fwd<N, false, true> = training, <N, false, false> = render-only).
Measurement helper, not product code."""
import ctypes, json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantsplat_amd import _lib
from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
from instantsplat_amd.gaussian_renderer import render
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import RunAhead, setup_training

dev = torch.device("cuda:0")
st = setup_training(syn_pointmap(3, 256, 256, 512, 512, seed=0), dev)
ra = RunAhead(st, window=10)
for _ in range(200):
    ra.step()
ra.flush()
if ra.trainer is not None:
    ra.trainer.close()
BinningPolicy.reset("exact")
g, L = st.gaussians, _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def frames(no_grad):
    imgs, t = [], 0.0
    for i in range(N * len(st.cameras) + 3):
        cam = st.cameras[i % len(st.cameras)]
        if i == 3:
            torch.cuda.synchronize()
            L.mi355gs_profile_set_period(1)
            L.mi355gs_profile_begin()
            t = time.perf_counter()
        if no_grad:
            with torch.no_grad():
                img = render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(cam.uid))["render"]
        else:
            img = render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(cam.uid))["render"].detach()
        if i < 3:
            imgs.append(img.clone())
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / (N * len(st.cameras))
    tot, n = ctypes.c_double(), ctypes.c_int()
    L.mi355gs_profile_read(2 if no_grad else 0, ctypes.byref(tot), ctypes.byref(n))
    L.mi355gs_profile_end()
    return imgs, 1e3 * tot.value / max(n.value, 1), n.value, 1e3 * wall


a_imgs, a_us, a_n, a_wall = frames(no_grad=False)
b_imgs, b_us, b_n, b_wall = frames(no_grad=True)
assert all(torch.equal(x, y) for x, y in zip(a_imgs, b_imgs)), "the two instantiations must give the same image"
print(json.dumps({"what": "k_composite_fwd at C3 (512^2, 196,608 Gaussians, iteration 200), same frames", "frames_each": a_n,
                  "training_instantiation_us": a_us, "render_only_instantiation_us": b_us,
                  "render_call_wall_ms_training": a_wall, "render_call_wall_ms_render_only": b_wall, "images_bit_identical": True}))
