"""Rate of the drop-in reference loop (and of the one-call synced loop) on the C3 scene, N iterations after a warm-up — for A/B of
process-level settings (environment variables of the HIP / HSA runtime have to be in place before the process starts).
Measurement helper, not product code."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantsplat_amd import train
from instantsplat_amd.arguments import OptimizationParams
from instantsplat_amd.synthetic import syn_pointmap
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda:0")
from instantsplat_amd.launch import pin_mode, pin_rank_to_cpu_slice
if pin_mode() != "off":   # what bench.py and the launcher do (MI355GS_PIN = node | compact | off)
    pin_rank_to_cpu_slice(0, 1, device_of_rank=lambda r: 0, compact=pin_mode() == "compact")
import gc
if os.environ.get("GS_GC") == "off":
    gc.disable()
elif os.environ.get("GS_GC") == "freeze":
    gc.collect(); gc.freeze()
out = []
for name, kw in (("drop-in", {}), ("drop-in train.py loss", {"fused_loss": False}), ("one-call synced", {"fused_step": True})):
    st = train.setup_training(syn_pointmap(3, 256, 256, 512, 512, seed=0), dev, opt=OptimizationParams(iterations=10 ** 9, pp_optimizer=True, optim_pose=True))
    st.gaussians.oneupSHdegree = lambda: None
    for _ in range(200): train.train_iteration(st, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N): train.train_iteration(st, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    train.release_trainer(st)
    out.append("%s %.0f" % (name, N / dt))
print(" | ".join(out), flush=True)
