"""Same-box A/B of the one-call synced loop's wait: an event behind every iteration vs polling the pinned result words.
Alternates the two on fresh states, N iterations each, three passes.   Measurement helper, not product code."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantsplat_amd import train
from instantsplat_amd.arguments import OptimizationParams
from instantsplat_amd.synthetic import syn_pointmap
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda:0")
def run(with_event):
    train.WAIT_WITH_EVENT = with_event
    st = train.setup_training(syn_pointmap(3, 256, 256, 512, 512, seed=0), dev, opt=OptimizationParams(iterations=10 ** 9, pp_optimizer=True, optim_pose=True))
    st.gaussians.oneupSHdegree = lambda: None
    for _ in range(200): train.train_iteration(st, fused_step=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N): train.train_iteration(st, fused_step=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    train.release_trainer(st)
    return N / dt
for rep in range(3):
    for name, ev in (("event", True), ("poll", False)):
        print("pass %d  %-6s %7.1f it/s" % (rep, name, run(ev)), flush=True)
