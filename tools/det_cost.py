"""What the deterministic-backward mode costs (include/mi355gs.h, mi355gs_tune_deterministic): C3 (196,608 Gaussians, 512^2) from
the state of iteration 200, N iterations of the one-call loop and of the drop-in loop (train.py loss as written) in both modes,
wall clock; and two deterministic runs compared bit for bit.  Under `bash tools/prof.sh r05_det_c3 python tools/det_cost.py` the
kernel stats show the mode's own kernels (k_det_area, k_scan_*, k_det_rowidx, the memset, k_composite_bwd<1, false, true>,
k_det_gather).  Measurement helper, not product code."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import instantsplat_amd.diff_gaussian_rasterization as dgr
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import RunAhead, release_trainer, setup_training, train_iteration
from instantsplat_amd.arguments import OptimizationParams

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
scene = syn_pointmap(3, 256, 256, 512, 512, seed=0)
opt = OptimizationParams(iterations=10 ** 9, pp_optimizer=True, optim_pose=True)
names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P")


def run(det, fused_step):
    dgr.set_deterministic(det)
    st = setup_training(scene, dev, opt=opt)
    st.gaussians.oneupSHdegree = lambda: None
    ra = RunAhead(st, window=10)
    for _ in range(200):
        ra.step()
    ra.flush()
    if ra.trainer is not None:
        ra.trainer.close()
    dgr.BinningPolicy.reset("exact")
    step = (lambda: train_iteration(st, fused_step=True)) if fused_step else (lambda: train_iteration(st, fused_loss=False))
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        last = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    release_trainer(st)
    dgr.BinningPolicy.reset("exact")
    dgr.set_deterministic(False)
    return N / dt, last, [getattr(st.gaussians, n).detach().clone() for n in names]


out = {"what": f"C3, iterations 221..{220 + N} of training from seed 0, wall clock", "iterations": N}
for fused_step, label in ((True, "one_call_synced"), (False, "dropin_train_py_loss")):
    a, _, _ = run(False, fused_step)
    b, lb, pb = run(True, fused_step)
    c, lc, pc = run(True, fused_step)
    same = lb == lc and all(torch.equal(x, y) for x, y in zip(pb, pc))
    out[label] = {"default_iters_per_sec": a, "deterministic_iters_per_sec": [b, c], "deterministic_over_default": 0.5 * (b + c) / a,
                  "two_deterministic_runs_bit_identical": bool(same), "last_loss": [lb, lc]}
print(json.dumps(out))
