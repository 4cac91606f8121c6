"""Random-shape fuzz of the operators around the rasterizer against their oracles: fused SSIM (both paddings; the reference's own
utils/loss_utils.py ssim as oracle/ssim_ref.py), the recorded loss expression of train.py:171-176 (value, gradient, early item),
distCUDA2 (float64 k-d tree), and short training runs against the CPU trainer — sizes the fixed-size tests do not visit (frames
from 1 x 1 to ~700 x 700, not multiples of the kernels' 32 x 32 tiles; clouds from 4 points to 30 k with duplicates).
python tools/fuzz_ops.py <seed> <cases> [gpu]   (default: the emulated kernels on the CPU).  Test tooling, not product code."""
import os, sys, time, random, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from instantsplat_amd import _lib
ON_GPU = len(sys.argv) > 3 and sys.argv[3] == "gpu"
if not ON_GPU:
    _lib._use_library_for_testing(os.environ.get("MI355GS_EMU_LIB") or os.path.join(ROOT, "tests", "emu", "libmi355gs_emu.so"))
from tests import ops_util as U
dev = torch.device("cuda:0" if ON_GPU else "cpu")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
big = 700 if ON_GPU else 90
bad, t0, counts = 0, time.time(), {}
for i in range(n_cases):
    kind = rng.choice(["ssim", "ssim", "ssim", "loss", "knn", "knn", "train"])
    try:
        if kind == "ssim":
            pad = rng.choice(["same", "valid"])
            lo = 11 if pad == "valid" else 1
            H = rng.choice([lo, lo + 1, 31, 32, 33, 63, 64, 65, rng.randrange(lo, big)]); W = rng.choice([lo, lo + 2, 31, 32, 33, 64, 97, rng.randrange(lo, big)])
            cfg = dict(kind=kind, H=H, W=W, padding=pad, seed=rng.randrange(1000))
            U.check_ssim_random(dev, H, W, seed=cfg["seed"], padding=pad)
        elif kind == "loss":
            H, W = rng.randrange(1, big), rng.randrange(1, big)
            cfg = dict(kind=kind, H=H, W=W)
            U.check_lazy_loss_expression(dev, H=H, W=W)
        elif kind == "knn":
            n = rng.choice([4, 5, 63, 64, 65, 1000, rng.randrange(4, 30000 if ON_GPU else 3000)])
            cfg = dict(kind=kind, n=n, seed=rng.randrange(1000), duplicates=rng.random() < 0.5)
            U.check_knn(dev, n, seed=cfg["seed"], duplicates=cfg["duplicates"])
        else:
            Wm = rng.randrange(4, 28 if ON_GPU else 12); W = rng.choice([16, 24, 33, 48, 70, 96] if ON_GPU else [16, 24, 33])
            cfg = dict(kind=kind, iters=3, Wm=Wm, W=W, fused_step=rng.random() < 0.5)
            U.check_train_matches_cpu_oracle(dev, 3, Wm=Wm, W=W, fused_step=cfg["fused_step"])
        counts[kind] = counts.get(kind, 0) + 1
    except Exception as e:
        bad += 1
        print("FAIL", cfg, type(e).__name__, str(e)[:300], flush=True)
        traceback.print_exc(limit=3)
print("seed", sys.argv[1] if len(sys.argv) > 1 else 0, "cases", n_cases, counts, "failures", bad, "in %.0f s" % (time.time() - t0),
      "on", "cuda:0 (libmi355gs.so)" if ON_GPU else "the emulator", flush=True)
