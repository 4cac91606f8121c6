"""Per-wave timing of the forward composite kernel on the C3 scene (measurement build, `make -C instantsplat_amd/csrc probe`).

Loads lib/libmi355gs_probe.so in place of the product library, trains the C3 student for a few iterations, renders one view with
the probe buffer armed and prints where a tile's waves spend their time: total lifetime, time waiting at the two workgroup
barriers of each 512-record batch (staging), time in the hit walks, hits per wave, and the per-CU picture.
"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantsplat_amd import _lib
PROBE = os.path.join(ROOT, "instantsplat_amd", "lib", "libmi355gs_probe.so")
_lib._use_library_for_testing(PROBE)
L = _lib.lib()
from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
from instantsplat_amd.gaussian_renderer import render
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import RunAhead, setup_training

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
st = setup_training(syn_pointmap(3, 256, 256, 512, 512, seed=0), dev)
ra = RunAhead(st, window=10)
for _ in range(iters):
    ra.step()
ra.flush()
BinningPolicy.reset("exact")
torch.cuda.synchronize()
T = 1024
buf = torch.zeros(T * 4, 8, dtype=torch.int64, device=dev)
raw = ctypes.CDLL(PROBE)
raw.mi355gs_probe_set.argtypes = [ctypes.c_void_p, ctypes.c_uint]
cam = st.cameras[0]
with torch.no_grad():
    for _ in range(3):
        render(cam, st.gaussians, st.pipe, st.background, camera_pose=st.gaussians.get_RT(cam.uid))
    torch.cuda.synchronize()
    assert raw.mi355gs_probe_set(ctypes.c_void_p(buf.data_ptr()), T * 4) == 0
    render(cam, st.gaussians, st.pipe, st.background, camera_pose=st.gaussians.get_RT(cam.uid))
    torch.cuda.synchronize()
    raw.mi355gs_probe_set(ctypes.c_void_p(0), 0)
b = buf.cpu().numpy().astype(np.float64)
if os.environ.get("GS_PROBE_DUMP"):   # raw rows (t0, t1, hits, wait, groups, instances, physical CU, walk ticks) per (tile, quadrant) for offline what-ifs
    np.save(os.environ["GS_PROBE_DUMP"], buf.cpu().numpy())
t0, t1, hits, wait, groups, inst, cu, walk = [b[:, i] for i in range(8)]
ok = t1 > 0
tick = 0.01  # wall_clock64: 100 MHz -> us
k0 = t0[ok].min()
life = (t1 - t0) * tick
print("waves recorded %d of %d; kernel span (first wave start -> last wave end) %.1f us" % (ok.sum(), len(ok), (t1[ok].max() - k0) * tick))
print("wave start spread: p50 %.1f p99 %.1f max %.1f us after the first" % tuple(np.percentile((t0[ok] - k0) * tick, [50, 99, 100])))
def pct(x): return "mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % (x.mean(), *np.percentile(x, [50, 90, 99, 100]))
print("wave lifetime  [us]: " + pct(life[ok]))
print("barrier wait   [us]: " + pct(wait[ok] * tick))
print("hit walks      [us]: " + pct(walk[ok] * tick))
print("other (cull, staging issue, boundary stores) [us]: " + pct((life - wait * tick - walk * tick)[ok]))
print("hits per wave      : " + pct(hits[ok]))
print("instances per tile : " + pct(inst[ok][::4]))
print("walk ns per hit (waves with >= 50 hits): " + pct((walk * tick * 1e3 / np.maximum(hits, 1))[ok & (hits >= 50)]))
# per tile: heaviest quadrant vs mean quadrant
h4 = hits.reshape(-1, 4)
print("quadrant imbalance: mean over tiles of max(hits)/mean(hits) = %.2f" % np.mean(h4.max(1) / np.maximum(h4.mean(1), 1)))
# per CU
cus = np.unique(cu[ok])
end_cu = np.array([(t1[ok & (cu == c)].max() - k0) * tick for c in cus])
hits_cu = np.array([hits[ok & (cu == c)].sum() for c in cus])
print("CUs used %d; CU finish time [us]: %s; hits per CU: %s; corr(hits, finish) %.2f" % (len(cus), pct(end_cu), pct(hits_cu), np.corrcoef(hits_cu, end_cu)[0, 1]))
w_cu = np.array([(ok & (cu == c)).sum() for c in cus])
print("waves per CU: " + pct(w_cu.astype(float)))
# the slowest waves
order = np.argsort(-(t1 - k0))[:8]
for i in order:
    print("  late wave: tile %4d q%d  start %.1f end %.1f us  hits %4d  inst %5d  wait %.1f walk %.1f" % (i // 4, i % 4, (t0[i] - k0) * tick, (t1[i] - k0) * tick, hits[i], inst[i], wait[i] * tick, walk[i] * tick))
# per-quadrant bias and the per-SIMD picture (wave w of a workgroup runs on SIMD w of its CU)
print("mean hits per quadrant (q0 top-left, q1 top-right, q2 bottom-left, q3 bottom-right): " + " ".join("%.0f" % h4[:, q].mean() for q in range(4)))
qi = np.arange(len(hits)) % 4
simd_hits = np.array([[hits[ok & (cu == c) & (qi == q)].sum() for q in range(4)] for c in cus])
simd_end = np.array([[((t1 - k0) * tick)[ok & (cu == c) & (qi == q)].max() for q in range(4)] for c in cus])
print("hits per SIMD: " + pct(simd_hits.ravel()) + "; max/mean %.2f" % (simd_hits.max() / simd_hits.mean()))
print("SIMD finish [us]: " + pct(simd_end.ravel()) + "; corr(hits, finish) %.2f" % np.corrcoef(simd_hits.ravel(), simd_end.ravel())[0, 1])
print("within a CU: mean of max(SIMD hits)/mean(SIMD hits) = %.2f" % np.mean(simd_hits.max(1) / simd_hits.mean(1)))
# where does the dispatcher put consecutive workgroups?  (workgroup index = rank of the tile by instance count, descending)
tile_inst = inst[::4]
tile_cu = cu[::4].astype(int)
rank_of = np.argsort(-tile_inst, kind="stable")       # rank -> tile (ties may differ from the device's counting sort)
cu_by_rank = tile_cu[rank_of]
print("CU of the first 24 workgroups: " + " ".join(str(c) for c in cu_by_rank[:24]))
for period in (8, 32, 64, 128, 256):
    same = np.mean(cu_by_rank[:-period] == cu_by_rank[period:]) if len(cu_by_rank) > period else float("nan")
    print("  fraction of workgroups on the same CU as workgroup index - %d: %.2f" % (period, same))
xcc = cu_by_rank >> 6
print("XCC of the first 24 workgroups: " + " ".join(str(c) for c in xcc[:24]))
# is the placement the same from launch to launch?  (same frame rendered again, also during a training run)
def placement():
    with torch.no_grad():
        buf.zero_()
        assert raw.mi355gs_probe_set(ctypes.c_void_p(buf.data_ptr()), T * 4) == 0
        render(cam, st.gaussians, st.pipe, st.background, camera_pose=st.gaussians.get_RT(cam.uid))
        torch.cuda.synchronize()
        raw.mi355gs_probe_set(ctypes.c_void_p(0), 0)
    return buf.cpu().numpy()[::4, 6].astype(int)
runs = [placement() for _ in range(4)]
ra2 = RunAhead(st, window=10)
for _ in range(20):
    ra2.step()
ra2.flush(); BinningPolicy.reset("exact"); torch.cuda.synchronize()
runs.append(placement())
base = runs[0][rank_of]
for i, r in enumerate(runs[1:], 1):
    print("placement of run %d vs run 0 (by workgroup index): %.3f of the workgroups on the same CU" % (i, np.mean(r[rank_of] == base)))
print("workgroups per CU in run 0: " + pct(np.bincount(runs[0], minlength=512)[np.bincount(runs[0], minlength=512) > 0].astype(float)))
