"""Soak of the drop-in loop (train.py's loss lines as written, loss.item() from the pinned word) across the SH-degree steps of the
reference's schedule (train.py:148-149: every 1000 iterations): 3100 iterations of C3 through training(run_ahead=False,
fused_loss=False), then the same 1100 iterations twice in the deterministic mode (bit-identical?), then the one-call loop.
Measurement / robustness helper, not product code."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import instantsplat_amd.diff_gaussian_rasterization as dgr
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import training

dev = torch.device("cuda:0")
scene = syn_pointmap(3, 256, 256, 512, 512, seed=0)
names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation", "P")
out = {}
r = training(scene, dev, iterations=3100, run_ahead=False, fused_loss=False)
out["dropin_3100_iterations"] = dict(iters_per_sec=r["iters_per_sec"], psnr_before=r["psnr_before"], psnr_after=r["psnr_after"], last_loss=r["last_loss"],
                                     sh_degree=int(r["state"].gaussians.active_sh_degree), f_rest_absmax=float(r["state"].gaussians._features_rest.abs().max()))
dgr.BinningPolicy.reset("exact")
r = training(scene, dev, iterations=3100, run_ahead=True)
out["one_call_3100_iterations"] = dict(iters_per_sec=r["iters_per_sec"], psnr_after=r["psnr_after"], last_loss=r["last_loss"], sh_degree=int(r["state"].gaussians.active_sh_degree))
dgr.BinningPolicy.reset("exact")
dgr.set_deterministic(True)
runs = []
for _ in range(2):
    r = training(scene, dev, iterations=1100, run_ahead=False, fused_loss=False)
    runs.append((r["last_loss"], r["psnr_after"], [getattr(r["state"].gaussians, n).detach().clone() for n in names]))
    dgr.BinningPolicy.reset("exact")
dgr.set_deterministic(False)
out["deterministic_1100_iterations_across_the_first_sh_step"] = dict(psnr_after=[runs[0][1], runs[1][1]], last_loss=[runs[0][0], runs[1][0]],
                                                                     bit_identical=bool(runs[0][0] == runs[1][0] and all(torch.equal(a, b) for a, b in zip(runs[0][2], runs[1][2]))))
print(json.dumps(out))
