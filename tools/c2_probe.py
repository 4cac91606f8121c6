"""Why does BASELINE configs[1] (C2: 50k random Gaussians, 512^2, forward only) read 0.48 ms per frame inside bench.py and 0.059 ms
standalone (tools/configs.py)?  Times the same operator call in a fresh process, after a C3 state has trained, with / without the
CPU pinning, and with the caching allocator emptied.  Measurement helper, not product code."""
import math, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantsplat_amd.synthetic import syn_blob, syn_pointmap
from instantsplat_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, BinningPolicy
dev = torch.device("cuda:0")


def c2(tag, n=200):
    sc = syn_blob(50000, 512, 512, seed=0)
    cam = sc.camera
    stg = GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), sc.bg.to(dev), 1.0,
                                        torch.eye(4, device=dev), cam.projection_matrix.to(dev), 3, torch.zeros(3, device=dev), False, False)
    a = dict(means3D=sc.means3D.to(dev), means2D=torch.zeros(50000, 3, device=dev), opacities=torch.sigmoid(sc.opacity_logit).to(dev),
             shs=sc.shs.to(dev), scales=torch.exp(sc.scaling_logit).to(dev), rotations=sc.rotation.to(dev))
    r = GaussianRasterizer(stg)
    with torch.no_grad():
        for _ in range(5):
            r(**a)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            r(**a)
        torch.cuda.synchronize()
        print(f"{tag}: {1e3 * (time.perf_counter() - t) / n:.4f} ms/frame", flush=True)


c2("fresh process")
from instantsplat_amd.train import setup_training, train_iteration, release_trainer
st = setup_training(syn_pointmap(3, 256, 256, 512, 512, seed=0), dev)
for _ in range(30):
    train_iteration(st, fused_loss=False)
c2("after 30 drop-in iterations of C3")
for _ in range(30):
    train_iteration(st, fused_step=True)
release_trainer(st)
c2("after 30 one-call iterations")
BinningPolicy.reset("bounded"); BinningPolicy.reset("exact")
c2("after BinningPolicy bounded -> exact")
del st
torch.cuda.empty_cache()
c2("after empty_cache")
from instantsplat_amd.launch import pin_rank_to_cpu_slice
pin_rank_to_cpu_slice(0, 1, device_of_rank=lambda r: 0)
c2("after pinning to the GPU's NUMA node")
