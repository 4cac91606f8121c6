"""Distribution of the trained-state parity ratios of tests/test_baseline_sizes_gpu.py over repeated runs (VERDICT r3 #7b).
ratio = (device error vs the fp64 oracle) / (fp32 oracle's own error vs the fp64 oracle), per tensor and view; the trained
state differs from run to run (float-atomic order), and so does which (pixel, Gaussian) pairs sit on a threshold.
usage (GPU box): python tools/trained_state_ratios.py [runs_c3] [runs_c4]      Measurement helper, not product code."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GS_CALIBRATE"] = "1"   # tests/ops_util.bound records instead of asserting
import torch
from instantsplat_amd import _lib
from tests import test_baseline_sizes_gpu as T

_lib._use_library_for_testing(None)
gpu = torch.device("cuda:0")
rows = collections.defaultdict(list)
cur = {}


def rec_grad(pre, k, dut, c32, c64, factor):
    full, robust = T._grad_errors(dut, c64)
    full_ref, robust_ref = T._grad_errors(c32, c64)
    cur.setdefault("full", []).append(full / max(full_ref, 1e-3 / factor))       # floors as in the test's limits
    cur.setdefault("robust", []).append(robust / max(robust_ref, 1e-4 / factor))


def rec_image(pre, dut, c32, c64, factor):
    d = (dut.detach().double().cpu() - c64).abs()
    d_ref = (c32.detach().double() - c64).abs()
    frac, frac_ref = float((d > 1e-4).double().mean()), float((d_ref > 1e-4).double().mean())
    cur.setdefault("image", []).append(frac / max(frac_ref, 2e-4 / factor))
    cur.setdefault("image_max", []).append(float(d.max()))


T._check_grad, T._check_image = rec_grad, rec_image
T.bound = lambda *a, **k: None
for tag, args, runs in (("C3 after 40 iterations, 3 views", ("C3", 3, 256, 512, 512, (0, 1, 2), 40), int(sys.argv[1]) if len(sys.argv) > 1 else 30),
                        ("C4 after 12 iterations, view 5", ("C4", 12, 288, 1920, 1080, (5,), 12), int(sys.argv[2]) if len(sys.argv) > 2 else 6)):
    for r in range(runs):
        cur.clear()
        T._compare_views_with_cpu_oracle(gpu, *args)
        for k, v in cur.items():
            rows[(tag, k)].append(max(v))
    for k in ("full", "robust", "image", "image_max"):
        v = sorted(rows[(tag, k)])
        print("%-34s %-10s runs %2d  min %.3g  median %.3g  p90 %.3g  max %.3g   all: %s" % (
            tag, k, len(v), v[0], v[len(v) // 2], v[int(0.9 * (len(v) - 1))], v[-1], " ".join("%.2f" % x for x in v)), flush=True)
print("ratio = worst tensor / view of a run; `full` = relative L2 of a gradient tensor against fp64 in units of the fp32 oracle's, "
      "`robust` = the same with the 64 worst Gaussians set aside, `image` = fraction of values off by > 1e-4 in the same units, "
      "image_max = largest |device - fp64| of the image (limit 5e-3)")
