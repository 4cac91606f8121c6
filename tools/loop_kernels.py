"""Which kernels does ONE iteration of a loop launch?  Runs N iterations of one of the bench's loops on the C3 scene (after an
untimed warm-up) — run it under `tools/prof.sh <name> python tools/loop_kernels.py <loop> <N>` and divide the calls by N.
loop: dropin_train_py (train.py's loss lines as written: bench.py's headline) | dropin (the loss as one fused call) | dropin_eager
(lazy_loss off) | dropin_torch_l1 | one_call | run_ahead.   Measurement helper, not product code."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantsplat_amd.arguments import OptimizationParams
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import RunAhead, release_trainer, setup_training, train_iteration

loop, N = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda:0")
st = setup_training(syn_pointmap(3, 256, 256, 512, 512, seed=0), dev, opt=OptimizationParams(iterations=10 ** 9, pp_optimizer=True, optim_pose=True))
st.gaussians.oneupSHdegree = lambda: None
if loop == "run_ahead":
    ra = RunAhead(st, window=10)
    step = ra.step
else:
    if loop == "dropin_eager":
        from instantsplat_amd import lazy_loss
        lazy_loss.ENABLED = False
    step = {"dropin": lambda: train_iteration(st), "dropin_train_py": lambda: train_iteration(st, fused_loss=False),
            "dropin_eager": lambda: train_iteration(st, fused_loss=False), "dropin_torch_l1": lambda: train_iteration(st, fused_loss="torch"),
            "one_call": lambda: train_iteration(st, fused_step=True)}[loop]
for _ in range(N):
    step()
torch.cuda.synchronize()
print("ran", 2 * 0 + N, "iterations of", loop)
