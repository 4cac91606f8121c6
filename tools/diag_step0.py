"""Element-level look at the first optimizer step of the reference-trajectory test on the device."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import ops_util
from instantsplat_amd.train import train_iteration, _forward_backward_step, _optimizer_step
dev = "cuda"
G, st, _ = ops_util._reference_loop_start(dev, "loop")
g = st.gaussians
T = lambda k: torch.from_numpy(G[k])
for n in ("_xyz", "_features_dc", "_opacity", "_scaling", "_rotation"):
    d = (getattr(g, n).detach().cpu() - T("loop_iter_params" + n)[0]).abs()
    print("init", n, float(d.max()))
loss = _forward_backward_step(st, True)
print("loss", float(loss))
gr = g._features_dc.grad
print("grad f_dc: shape", tuple(gr.shape), "stride", gr.stride(), "contig", gr.is_contiguous(), "ptr%16", gr.data_ptr() % 16, "storage_offset", gr.storage_offset())
gd = gr.detach().cpu().flatten(); gg = T("loop_iter_grads_features_dc")[0].flatten()
print("grad diff max", float((gd - gg).abs().max()), "max", float(gg.abs().max()))
p0 = g._features_dc.detach().cpu().flatten().clone()
grad_before = gr.detach().clone()
_optimizer_step(st)
torch.cuda.synchronize()
p1 = g._features_dc.detach().cpu().flatten()
ref1 = T("loop_iter_params_features_dc")[1].flatten()
d = (p1 - ref1).abs()
bad = torch.nonzero(d > 1e-4).flatten()
print("after step: bad elements", bad.numel(), "of", d.numel())
for i in bad.tolist()[:40]:
    print("  el %3d  p0 %.6f p1 %.6f ref1 %.6f  moved %.6f ref moved %.6f  g %.3e  gref %.3e" % (
        i, p0[i], p1[i], ref1[i], p1[i] - p0[i], ref1[i] - T("loop_iter_params_features_dc")[0].flatten()[i], gd[i], gg[i]))
s = g.optimizer.state[g._features_dc]
m = s["exp_avg"].detach().cpu().flatten(); v = s["exp_avg_sq"].detach().cpu().flatten()
print("m vs 0.1 g: max diff", float((m - 0.1 * gd).abs().max()), " v vs 0.001 g^2 rel", float(((v - 0.001 * gd * gd).abs() / (0.001 * gd * gd + 1e-30)).max()))
print("v min", float(v.min()), "g min abs", float(gd.abs().min()))
