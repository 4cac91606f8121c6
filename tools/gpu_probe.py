"""Ad-hoc GPU probe (not part of the product): time the raster stages at the BASELINE shapes."""
import math, sys, time, torch
sys.path.insert(0, '.'); sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from instantsplat_amd.synthetic import syn_blob
from instantsplat_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from tests.util import settings_for

dev = torch.device('cuda:0')
print(torch.cuda.get_device_name(0))
CASES = [(50000, 512, 512, 3, 'random'), (200000, 512, 512, 0, 'random'), (200000, 512, 512, 0, 'init'), (1000000, 1920, 1080, 0, 'random')]
if len(sys.argv) > 1: CASES = [CASES[int(a)] for a in sys.argv[1:]]
for (P, W, H, deg, op) in CASES:
    sc = syn_blob(P, W, H, seed=0, opacity=op)
    st = settings_for(sc.camera, deg, GaussianRasterizationSettings, sc.bg, device=dev)
    means = sc.means3D.to(dev).requires_grad_(True); sh = sc.shs.to(dev).requires_grad_(True)
    opl = sc.opacity_logit.to(dev).requires_grad_(True); scl = sc.scaling_logit.to(dev).requires_grad_(True); rot = sc.rotation.to(dev).requires_grad_(True)
    m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
    r = GaussianRasterizer(st)
    def fwd():
        return r(means3D=means, means2D=m2d, opacities=torch.sigmoid(opl), shs=sh, scales=torch.exp(scl), rotations=rot)
    for _ in range(3):
        c, radii = fwd(); c.sum().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 10
    for _ in range(n):
        with torch.no_grad(): c, radii = fwd()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(n):
        c, radii = fwd(); c.sum().backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"P={P} {W}x{H} deg={deg} op={op}: fwd {1e3*(t1-t0)/n:.3f} ms  fwd+bwd {1e3*(t2-t1)/n:.3f} ms  visible {int((radii>0).sum())}")
