"""Where does the device-vs-oracle gradient difference at 200k Gaussians / 512x512 come from?  Device (fp32), C oracle fp32
and C oracle fp64 on the same inputs: pairwise relative L2 per tensor, and how concentrated the squared error is."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantsplat_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from instantsplat_amd.synthetic import syn_blob
from oracle import gs_ref, raster_torch as rt
from tests.util import settings_for

P, W, H, deg = 200000, 512, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 0
gs_ref.lib().gsref_set_threads(32)
sc = syn_blob(P, W, H, seed=0, scale_mean=0.02)
torch.manual_seed(100)
wgt = torch.randn(3, H, W)
res = {}
for which in ("dut", "c32", "c64"):
    dt = torch.float64 if which == "c64" else torch.float32
    dev = torch.device("cuda:0") if which == "dut" else torch.device("cpu")
    lv = dict(means3D=sc.means3D, scaling=sc.scaling_logit, rot=sc.rotation, op=sc.opacity_logit, shs=sc.shs)
    lv = {k: v.clone().to(dt).to(dev).requires_grad_(True) for k, v in lv.items()}
    m2d = torch.zeros(P, 3, dtype=dt, device=dev, requires_grad=True)
    kw = dict(shs=lv["shs"], scales=torch.exp(lv["scaling"]), rotations=lv["rot"])
    if which == "dut":
        st = settings_for(sc.camera, deg, GaussianRasterizationSettings, torch.tensor([0.2, 0.5, 0.9]), device=dev)
        color, radii = GaussianRasterizer(st)(means3D=lv["means3D"], means2D=m2d, opacities=torch.sigmoid(lv["op"]), **kw)
    else:
        st = settings_for(sc.camera, deg, rt.RasterSettings, torch.tensor([0.2, 0.5, 0.9]))
        if which == "c64":
            st = rt.RasterSettings(*[(x.double() if isinstance(x, torch.Tensor) else x) for x in st])
        color, radii = gs_ref.rasterize(lv["means3D"], m2d, torch.sigmoid(lv["op"]), st, **kw)
    (color * wgt.to(dt).to(dev)).sum().backward()
    res[which] = dict(color=color.detach().cpu().double(), radii=radii.cpu(),
                      grads={**{k: v.grad.detach().cpu().double() for k, v in lv.items()}, "means2D": m2d.grad.detach().cpu().double()})
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-300))
for a, b in (("dut", "c32"), ("dut", "c64"), ("c32", "c64")):
    d = (res[a]["color"] - res[b]["color"]).abs()
    print("%s vs %s: image max %.2e  frac>1e-4 %.2e  frac>1e-5 %.2e  radii differ %d" % (a, b, float(d.max()), float((d > 1e-4).double().mean()),
          float((d > 1e-5).double().mean()), int((res[a]["radii"] != res[b]["radii"]).sum())))
    for k in res[a]["grads"]:
        ga, gb = res[a]["grads"][k], res[b]["grads"][k]
        e = ((ga - gb) ** 2).reshape(P, -1).sum(1)
        tot = float(e.sum())
        top = torch.sort(e, descending=True).values
        print("   grad %-8s rel-L2 %.2e   share of squared error in top 10/100/1000 Gaussians: %.2f %.2f %.2f   #Gaussians with rel err>1e-3: %d" % (
            k, rel(ga, gb), float(top[:10].sum()) / tot, float(top[:100].sum()) / tot, float(top[:1000].sum()) / tot,
            int((e.sqrt() > 1e-3 * (gb ** 2).reshape(P, -1).sum(1).sqrt() + 1e-12 * float(gb.abs().max())).sum())))
