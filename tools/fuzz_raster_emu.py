"""CPU-only fuzz of the rasterizer kernels (the unmodified .hip sources under the SIMT emulator of tests/emu) against the fp32 C oracle:
random Gaussian counts (1 .. 4000), image sizes (incl. sizes that are not multiples of the tile, and > 1024 tiles), SH degrees, scale
distributions from sub-pixel to larger than the image, opacities, scale modifiers, precomputed colours / covariances — the parity
criteria of tests/util.py::assert_raster_parity.  Round 5: every other case also runs in the deterministic-backward mode (same
criteria, and gradients within rounding of the default mode's) and takes its forward once more under torch.no_grad() — the
render-only stage 2 — which must give the same image and radii bit for bit.
python tools/fuzz_raster_emu.py <seed> <cases>.  Test tooling, not product code."""
import sys, time, random, traceback
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from instantsplat_amd import _lib
_lib._use_library_for_testing(__import__('os').environ.get('MI355GS_EMU_LIB') or __import__('os').path.join(sys.path[0], 'tests', 'emu', 'libmi355gs_emu.so'))
from tests.util import assert_raster_parity, relerr, run_blob_case
import instantsplat_amd.diff_gaussian_rasterization as dgr
dev = torch.device('cpu')
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
t0 = time.time()
for i in range(n_cases):
    big = rng.random() < 0.12
    P = rng.choice([1, 2, 7, 63, 64, 65, 200, 513, 900, 1500, 2500, 4000]) if not big else rng.choice([30, 200, 600])
    W = rng.choice([16, 17, 31, 33, 48, 64, 100, 130, 200]) if not big else rng.choice([520, 640, 300])
    H = rng.choice([16, 15, 33, 48, 70, 96, 150]) if not big else rng.choice([528, 400, 272])
    deg = rng.choice([0, 1, 2, 3]); sm = rng.choice([0.005, 0.01, 0.03, 0.08, 0.15, 0.3, 0.6, 1.2, 2.5])
    op = rng.choice(["random", "init"]); mod = rng.choice([1.0, 1.0, 0.6, 1.7]); seed = rng.randrange(1000)
    pc, pv = rng.random() < 0.15, rng.random() < 0.15
    cfg = dict(P=P, W=W, H=H, deg=deg, sm=sm, op=op, mod=mod, seed=seed, pc=pc, pv=pv)
    try:
        out = run_blob_case(dev, P, W, H, deg, scale_mean=sm, seed=seed, opacity=op, mod=mod, precomp_color=pc, precomp_cov=pv)
        assert_raster_parity(out)
        if i % 2 == 0:
            dgr.set_deterministic(True)
            try:
                det = run_blob_case(dev, P, W, H, deg, scale_mean=sm, seed=seed, opacity=op, mod=mod, precomp_color=pc, precomp_cov=pv)
            finally:
                dgr.set_deterministic(False)
            assert_raster_parity(det)
            assert torch.equal(det["dut"]["color"], out["dut"]["color"])
            for k, g in out["dut"]["grads"].items():
                assert relerr(det["dut"]["grads"][k], g) <= 1e-5 or float(g.abs().max()) == 0.0, ("deterministic mode", k, relerr(det["dut"]["grads"][k], g))
            with torch.no_grad():
                ro = run_blob_case(dev, P, W, H, deg, scale_mean=sm, seed=seed, opacity=op, mod=mod, precomp_color=pc, precomp_cov=pv, backward=False)
            assert torch.equal(ro["dut"]["color"], out["dut"]["color"]) and torch.equal(ro["dut"]["radii"], out["dut"]["radii"]), "render-only forward"
    except Exception as e:
        bad += 1
        print("FAIL", cfg, type(e).__name__, str(e)[:300], flush=True)
print("seed", sys.argv[1] if len(sys.argv) > 1 else 0, "cases", n_cases, "failures", bad, "in %.0f s" % (time.time() - t0), flush=True)
