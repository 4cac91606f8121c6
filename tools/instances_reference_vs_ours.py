"""How many (tile, Gaussian) instances the REFERENCE operator's binning holds for the frames bench.py times, against this library's:
the oracle (oracle/gs_ref.c) bins by the published 3-sigma rectangle, the library intersects it with the box around
{alpha >= 1/255} and drops tiles no pixel of which can pass the alpha test (csrc/preprocess.hip) — same image, same gradients,
fewer instances.  SURVEY 8(d) counts a kernel's algorithmic bytes per CONSUMED instance (R_eff = sum over tiles of the deepest
last contributor); this prints R and R_eff both ways for the C3 state after `iters` training iterations, and the composite
backward's algorithmic bytes in the reference's accounting.  usage (GPU box): python tools/instances_reference_vs_ours.py [iters]
Measurement helper (uses the oracle as a yardstick), not product code."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy, keep_last_frame, last_frame_stats, reference_instance_count
from instantsplat_amd.gaussian_renderer import render
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import RunAhead, setup_training
from oracle import gs_ref
from oracle.raster_torch import RasterSettings
from oracle.train_ref import _pose_to_w2c, _quadmul

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
gs_ref.lib().gsref_set_threads(32)
st = setup_training(syn_pointmap(3, 256, 256, 512, 512, seed=0), dev)
g = st.gaussians
ra = RunAhead(st, window=10)
for _ in range(iters):
    ra.step()
ra.flush(); torch.cuda.synchronize(); BinningPolicy.reset("exact")
keep_last_frame(True)
tot = dict(R=0, Reff=0, Rref=0, Reffref=0)
W = H = 512
with torch.no_grad():
    for cam in st.cameras:
        pose = g.get_RT(cam.uid)
        out = render(cam, g, st.pipe, st.background, camera_pose=pose)
        r, reff = last_frame_stats()
        pc = pose.detach().cpu()
        Rm, t = _pose_to_w2c(pc)
        means = g._xyz.detach().cpu() @ Rm.t() + t
        rots = _quadmul(pc[:4], g._rotation.detach().cpu())
        s = RasterSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), torch.zeros(3), 1.0, torch.eye(4),
                           cam.projection_matrix.cpu(), g.active_sh_degree, torch.zeros(3), False, False)
        shs = torch.cat([g._features_dc.detach().cpu(), g._features_rest.detach().cpu()], dim=1)
        _, _, ctx = gs_ref.forward(means, torch.sigmoid(g._opacity.detach().cpu()).reshape(-1), s, shs=shs,
                                   scales=torch.exp(g._scaling.detach().cpu()), rotations=rots)
        aux = ctx.aux(W, H)
        nc = aux["n_contrib"].reshape(H // 16, 16, W // 16, 16).permute(0, 2, 1, 3).reshape(-1, 256)
        rref, reffref = ctx.num_rendered, int(nc.max(dim=1).values.sum())
        counted = reference_instance_count(means.to(dev), cam.projection_matrix, out["radii"], W, H)   # what bench.py reports: no oracle
        print("view %d: this library R = %d, R_eff = %d | reference binning (oracle) R = %d, R_eff = %d | ratio R %.3f  R_eff %.3f | "
              "reference_instance_count() = %d (%+d against the oracle)" % (cam.uid, r, reff, rref, reffref, rref / r, reffref / reff, counted, counted - rref), flush=True)
        tot["R"] += r; tot["Reff"] += reff; tot["Rref"] += rref; tot["Reffref"] += reffref
n = len(st.cameras)
ours, ref = 112 * tot["Reff"] / n + 20 * W * H, 112 * tot["Reffref"] / n + 20 * W * H
print("C3 after %d iterations, mean of %d views: R %.0f -> reference %.0f (x %.3f); R_eff %.0f -> reference %.0f (x %.3f)" % (
    iters, n, tot["R"] / n, tot["Rref"] / n, tot["Rref"] / tot["R"], tot["Reff"] / n, tot["Reffref"] / n, tot["Reffref"] / tot["Reff"]))
print("composite backward, algorithmic bytes per launch (112 B x R_eff + 20 B x W H): %.1f MB on this library's lists, %.1f MB on the reference's" % (ours / 1e6, ref / 1e6))
