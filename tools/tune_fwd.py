"""Forward composite time on C3 as a function of the persistent workgroups per CU (mi355gs_tune_fwd_workgroups_per_cu)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantsplat_amd import _lib
from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import RunAhead, setup_training
from instantsplat_amd.arguments import OptimizationParams
L = _lib.lib()
dev = torch.device("cuda:0")
st = setup_training(syn_pointmap(3, 256, 256, 512, 512, seed=0), dev, opt=OptimizationParams(iterations=100000, pp_optimizer=True, optim_pose=True))
ra = RunAhead(st, window=10)
for _ in range(100):
    ra.step()
ra.flush()
for k in (0, 6, 4, 3, 2, 1, 0):
    L.mi355gs_tune_fwd_workgroups_per_cu(k)
    for _ in range(20):
        ra.step()
    ra.flush(); torch.cuda.synchronize()
    L.mi355gs_profile_begin()
    for _ in range(100):
        ra.step()
    ra.flush(); torch.cuda.synchronize()
    ms, n = ctypes.c_double(), ctypes.c_int()
    out = []
    for kind in (0, 1):
        L.mi355gs_profile_read(kind, ctypes.byref(ms), ctypes.byref(n))
        out.append(1e3 * ms.value / max(n.value, 1))
    L.mi355gs_profile_end()
    print("workgroups per CU %d: composite fwd %.1f us, bwd %.1f us" % (k, out[0], out[1]))
L.mi355gs_tune_fwd_workgroups_per_cu(0)
BinningPolicy.reset("exact")
