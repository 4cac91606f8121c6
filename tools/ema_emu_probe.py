"""Ad-hoc (CPU only): how much the 23-iteration loss EMA of tests/ops_util.py::check_run_ahead_equals_sync_loop moves when
nothing but floating-point contraction changes.  Usage:
    bash tests/emu/build_emu.sh                                   # emulated library, no FMA contraction on x86
    (same g++ command line plus -mfma, objects linked into /tmp/emufma/libemu_fma.so)
    python tools/ema_emu_probe.py tests/emu/libmi355gs_emu.so     # -> 0.0077158 (op-by-op) 0.0077158 (one-call)
    python tools/ema_emu_probe.py /tmp/emufma/libemu_fma.so       # -> 0.0077474            0.0077474
Within one build the two loops agree to 1e-11; between the builds the EMA moves by 0.4 % — the same size as the
0.5-1.1 % between the two loops on the GPU (0.00773 vs 0.00767), whose kernels are contracted differently by hipcc."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantsplat_amd import _lib
_lib._use_library_for_testing(os.path.abspath(sys.argv[1]))
from instantsplat_amd.arguments import OptimizationParams
from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import RunAhead, setup_training, train_iteration
dev = torch.device("cpu")
sc = syn_pointmap(3, 20, 20, 48, 48, seed=7)
mk = lambda: setup_training(sc, dev, opt=OptimizationParams(iterations=1000, pp_optimizer=True, optim_pose=True))
out = []
for mode in ("sync", "ahead"):
    st = mk()
    if mode == "sync":
        ema = 0.0
        for _ in range(23): ema = 0.4 * train_iteration(st) + 0.6 * ema
    else:
        ra = RunAhead(st, window=5, fused_step=True)
        for _ in range(23): ra.step()
        ema = ra.flush()
    out.append(ema); BinningPolicy.reset("exact")
print(os.path.basename(sys.argv[1]), "%.7f %.7f rel %.2e" % (out[0], out[1], abs(out[0] - out[1]) / out[0]))
