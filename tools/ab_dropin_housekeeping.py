"""Same-box A/B of the drop-in loop's per-iteration housekeeping (round 4): the pose-row node (no select-backward fill + copy)
and the forward-owned accumulator buffer (no memset in front of the backward).  Alternates the settings on ONE state-size,
N iterations each, three passes.   Measurement helper, not product code."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantsplat_amd import _lib
from instantsplat_amd.arguments import OptimizationParams
from instantsplat_amd.scene import GaussianModel
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import setup_training, train_iteration

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda:0")
ext = _lib.compiled()
def run(pose_row, own_scratch):
    GaussianModel.POSE_ROW_NODE = pose_row
    ext.forward_owns_scratch(own_scratch)
    st = setup_training(syn_pointmap(3, 256, 256, 512, 512, seed=0), dev, opt=OptimizationParams(iterations=10 ** 9, pp_optimizer=True, optim_pose=True))
    st.gaussians.oneupSHdegree = lambda: None
    for _ in range(200): train_iteration(st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N): train_iteration(st)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return N / dt
for rep in range(3):
    for name, a, b in (("before (fill+copy, memset)", False, False), ("pose-row node only", True, False), ("both (shipped)", True, True)):
        print("pass %d  %-28s %7.1f it/s" % (rep, name, run(a, b)), flush=True)
