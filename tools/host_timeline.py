"""Where the host spends an iteration of the drop-in (reference-shaped) loop: wall-clock of every stage of
train_iteration(sync_loss=True) on the C3 scene, both reference read-backs kept.  Stages that contain a device read-back
(`size_and_render`: the instance count; `loss.item()`) include the wait for the GPU.
usage: python tools/host_timeline.py [iterations] [train_py|fused]     train_py (default): the loss lines of train.py:171-176 as
written (l1_loss + fused_ssim + scalar arithmetic: instantsplat_amd/lazy_loss.py); fused: the loss as one call"""
import os, sys, time
from collections import defaultdict
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import instantsplat_amd.train as T
import instantsplat_amd.fused as F
import instantsplat_amd.diff_gaussian_rasterization as D
import instantsplat_amd.fused_ssim as S
from instantsplat_amd.synthetic import syn_pointmap

ACC, CNT = defaultdict(float), defaultdict(int)
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            ACC[name] += time.perf_counter() - t; CNT[name] += 1
    return w

dev = torch.device("cuda:0")
st = T.setup_training(syn_pointmap(3, 256, 256, 512, 512, seed=0), dev)
g = st.gaussians
F._RenderPosed.forward = staticmethod(timed("  _RenderPosed.forward (incl. size_and_render)", F._RenderPosed.forward))
F._RenderPosed.backward = staticmethod(timed("  _RenderPosed.backward (autograd thread)", F._RenderPosed.backward))
D.size_and_render = timed("    size_and_render (count read-back + binning/composite enqueue)", D.size_and_render)
S._FusedL1SSIM.forward = staticmethod(timed("  _FusedL1SSIM.forward", S._FusedL1SSIM.forward))
S._FusedL1SSIM.backward = staticmethod(timed("  _FusedL1SSIM.backward (autograd thread)", S._FusedL1SSIM.backward))
import instantsplat_amd.lazy_loss as LZ
MODE = sys.argv[2] if len(sys.argv) > 2 else "train_py"
FUSED = True if MODE == "fused" else False
T.l1_loss = timed(" l1_loss() [the loss pair: L1 + SSIM in one pass]", T.l1_loss)
T.fused_ssim = timed(" fused_ssim() [the other half of the pair: no launch]", T.fused_ssim)
LZ.LazyScalar.backward = timed(" loss.backward() [LazyScalar: materialise + engine run in one compiled call]", LZ.LazyScalar.backward)
LZ.LazyScalar.item = timed(" loss.item() [LazyScalar] (waits for the GPU)", LZ.LazyScalar.item)
LZ.LazyScalar.detach = timed(" loss.detach() [LazyScalar]", LZ.LazyScalar.detach)
T.render = timed(" render()", T.render)
T.fused_l1_ssim_loss = timed(" fused_l1_ssim_loss()", T.fused_l1_ssim_loss)
g.update_learning_rate = timed(" update_learning_rate", g.update_learning_rate)
g.get_RT = timed(" get_RT", g.get_RT)
T._pick_camera = timed(" _pick_camera", T._pick_camera)
g.optimizer.step = timed(" optimizer.step", g.optimizer.step)
g.optimizer.zero_grad = timed(" optimizer.zero_grad", g.optimizer.zero_grad)
_bw = torch.Tensor.backward
torch.Tensor.backward = timed(" loss.backward()", _bw)
_item = torch.Tensor.item
torch.Tensor.item = timed(" loss.item() (waits for the GPU)", _item)
T._forward_backward_step = timed("_forward_backward_step", T._forward_backward_step)
T._optimizer_step = timed("_optimizer_step", T._optimizer_step)

from instantsplat_amd import _lib
if os.environ.get("GS_SINGLE_THREAD_AUTOGRAD") == "1":
    torch.autograd.set_multithreading_enabled(False)   # backward on the calling thread: no hand-off to the device thread
print("binding:", _lib.BINDING, "| multithreaded autograd:", torch.autograd.is_multithreading_enabled())
for _ in range(100):
    T.train_iteration(st, fused_loss=FUSED, sync_loss=True)
torch.cuda.synchronize(); ACC.clear(); CNT.clear()
ext = _lib.compiled()
if ext is not None:
    ext.host_times_us(True)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
t0 = time.perf_counter()
for _ in range(N):
    T.train_iteration(st, fused_loss=FUSED, sync_loss=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("drop-in loop (%s loss) with timers: %.1f us per iteration (%.0f it/s)" % ("train.py:171-176 as written" if not FUSED else "one fused call", dt / N * 1e6, N / dt))
for k, v in ACC.items():
    print("%-70s %7.1f us/iter  (%d calls)" % (k, v / N * 1e6, CNT[k] // N))
if ext is not None:
    names = ("RenderPosedFn.forward (C++, incl. the count wait)", "RenderPosedFn.backward (C++)", "L1SsimLossFn.forward (C++)",
             "L1SsimLossFn.backward (C++)", "AdamPlan.step (C++)", "  of the forward: wait for the instance count")
    for n_, v in zip(names, ext.host_times_us(False)):
        print("  %-68s %7.1f us/iter" % (n_, v / N))
