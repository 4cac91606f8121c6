"""Second look at the C2 slowdown (tools/c2_probe.py: 0.059 ms per frame in a fresh process, 0.45 after 30 drop-in iterations of C3,
0.06 again after 30 one-call iterations): per-frame wall times (is it every frame, or a few frames at the poll's 2 ms fallback?),
the binding's own host clocks, and the same frames through the ctypes binding (stream synchronise instead of the pinned-word poll).
Measurement helper, not product code."""
import math, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantsplat_amd import _lib
from instantsplat_amd.synthetic import syn_blob, syn_pointmap
from instantsplat_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
dev = torch.device("cuda:0")
sc = syn_blob(50000, 512, 512, seed=0)
cam = sc.camera
stg = GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), sc.bg.to(dev), 1.0,
                                    torch.eye(4, device=dev), cam.projection_matrix.to(dev), 3, torch.zeros(3, device=dev), False, False)
a = dict(means3D=sc.means3D.to(dev), means2D=torch.zeros(50000, 3, device=dev), opacities=torch.sigmoid(sc.opacity_logit).to(dev),
         shs=sc.shs.to(dev), scales=torch.exp(sc.scaling_logit).to(dev), rotations=sc.rotation.to(dev))
r = GaussianRasterizer(stg)
ext = _lib.compiled()


def c2(tag, n=300):
    with torch.no_grad():
        for _ in range(5):
            r(**a)
        torch.cuda.synchronize()
        if ext is not None:
            ext.host_times_us(True)
        ts = []
        t0 = time.perf_counter()
        for _ in range(n):
            t = time.perf_counter()
            r(**a)
            ts.append(time.perf_counter() - t)
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
        ts.sort()
        clocks = [round(x / n, 1) for x in ext.host_times_us(True)] if ext is not None else None
        print(f"{tag}: {1e3 * tot / n:.4f} ms/frame; per call us: min {1e6 * ts[0]:.0f} median {1e6 * ts[n // 2]:.0f} p90 {1e6 * ts[int(.9 * n)]:.0f} "
              f"p99 {1e6 * ts[int(.99 * n)]:.0f} max {1e6 * ts[-1]:.0f}; calls over 1 ms: {sum(t > 1e-3 for t in ts)}; host clocks us/frame {clocks}", flush=True)


c2("fresh process")
from instantsplat_amd.train import setup_training, train_iteration, release_trainer
st = setup_training(syn_pointmap(3, 256, 256, 512, 512, seed=0), dev)
for _ in range(30):
    train_iteration(st, fused_loss=True)
c2("after 30 drop-in iterations, fused loss")
for _ in range(30):
    train_iteration(st, fused_loss=False)
c2("after 30 drop-in iterations, loss as written")
import gc
gc.collect()
c2("after gc.collect()")
torch.cuda.synchronize(); time.sleep(1.0)
c2("after a second of idling")
from instantsplat_amd import lazy_loss
lazy_loss.forget()
c2("after lazy_loss.forget()")
for p in (st.gaussians._xyz, st.gaussians._features_dc, st.gaussians._features_rest, st.gaussians._opacity, st.gaussians._scaling, st.gaussians._rotation, st.gaussians.P):
    p.grad = None
c2("after dropping the gradients")
for _ in range(30):
    train_iteration(st, fused_step=True)
release_trainer(st)
c2("after 30 one-call iterations")
for _ in range(30):
    train_iteration(st, fused_loss=False)
os.environ["X"] = "1"
_lib.BINDING = "ctypes"
c2("after 30 more drop-in iterations, C2 through the ctypes binding")
_lib.BINDING = "compiled"
from instantsplat_amd.launch import pin_rank_to_cpu_slice
pin_rank_to_cpu_slice(0, 1, device_of_rank=lambda r: 0)
c2("compiled binding again, after pinning to the GPU's NUMA node")
for _ in range(30):
    train_iteration(st, fused_loss=False)
c2("pinned, after 30 more drop-in iterations")
