"""Per-wave timing of the backward composite kernel on the C3 scene (measurement build, `make -C instantsplat_amd/csrc probe`)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantsplat_amd import _lib
PROBE = os.environ.get("GS_PROBE_LIB") or os.path.join(ROOT, "instantsplat_amd", "lib", "libmi355gs_probe.so")
_lib._use_library_for_testing(PROBE)
from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
from instantsplat_amd.fused_ssim import fused_l1_ssim_loss
from instantsplat_amd.gaussian_renderer import render
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import RunAhead, setup_training
dev = torch.device("cuda:0")
st = setup_training(syn_pointmap(3, 256, 256, 512, 512, seed=0), dev)
ra = RunAhead(st, window=10)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    ra.step()
ra.flush(); BinningPolicy.reset("exact"); torch.cuda.synchronize()
ROWS = 4096 + 40000
buf = torch.zeros(ROWS, 8, dtype=torch.int64, device=dev)
raw = ctypes.CDLL(PROBE)
raw.mi355gs_probe_set.argtypes = [ctypes.c_void_p, ctypes.c_uint]
cam, g = st.cameras[0], st.gaussians
def fb():
    img = render(cam, g, st.pipe, st.background, camera_pose=g.get_RT(cam.uid))["render"]
    loss, _ = fused_l1_ssim_loss(img.unsqueeze(0), st.gt_images[cam.uid].unsqueeze(0), 0.2)
    loss.backward()
    for p in (g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation, g.P): p.grad = None
for _ in range(3): fb()
torch.cuda.synchronize()
assert raw.mi355gs_probe_set(ctypes.c_void_p(buf.data_ptr()), ROWS) == 0
fb(); torch.cuda.synchronize()
raw.mi355gs_probe_set(ctypes.c_void_p(0), 0)
b = buf.cpu().numpy().astype(np.float64)[4096:]
t0, t1, steps, pro, tile, seg, cu, blk = [b[:, i] for i in range(8)]
ok = t1 > 0
tick = 0.01
k0 = t0[ok].min()
span = (t1[ok].max() - k0) * tick
life = (t1 - t0)[ok] * tick
def pct(x): return "mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (x.mean(), *np.percentile(x, [50, 90, 99, 100]))
print("units with work %d; kernel span %.1f us; sum of wave lifetimes / (span x 1024 SIMDs) = %.2f waves per SIMD on average" % (ok.sum(), span, life.sum() / (span * 1024)))
print("wave lifetime [us]: " + pct(life)); print("prologue (loads, wave maxima, until the first chunk) [us]: " + pct(pro[ok] * tick))
print("steps per unit: " + pct(steps[ok])); print("ns per step (units with >= 20 steps): " + pct((life * 1e3 / np.maximum(steps[ok], 1))[steps[ok] >= 20]))
# occupancy over time
edges = np.linspace(0, span, 21)
s0, s1 = (t0[ok] - k0) * tick, (t1[ok] - k0) * tick
occ = [np.sum(np.clip(np.minimum(s1, edges[i + 1]) - np.maximum(s0, edges[i]), 0, None)) / (edges[i + 1] - edges[i]) / 1024 for i in range(20)]
print("resident working waves per SIMD over the kernel (20 slices): " + " ".join("%.1f" % o for o in occ))
print("start time of waves [us]: " + pct(s0))
