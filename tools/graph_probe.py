"""Launch structure of the one-call train step, measured: the same 12-launch iteration (a) issued eagerly from
the library call, (b) replayed from a captured hipGraph (one graph per training view, arguments frozen at capture time — so the
Adam bias corrections and learning rates do not advance: a timing experiment, not a training mode).
    python tools/graph_probe.py [steps]
"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instantsplat_amd.diff_gaussian_rasterization import BinningPolicy
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import RunAhead, setup_training
from instantsplat_amd.arguments import OptimizationParams

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
st = setup_training(syn_pointmap(3, 256, 256, 512, 512, seed=0), dev, opt=OptimizationParams(iterations=100000, pp_optimizer=True, optim_pose=True))
ra = RunAhead(st, window=10)
for _ in range(60):
    ra.step()
ra.flush()
tr = ra.trainer
slot = torch.zeros(1, device=dev)
order = [0, 1, 2]

def eager(n):
    for i in range(n):
        st.viewpoint_stack = [st.cameras[order[i % 3]]]
        tr.step(slot, verify_async=False)

eager(30)
torch.cuda.synchronize()
t0 = time.perf_counter(); eager(steps); torch.cuda.synchronize(); t_eager = (time.perf_counter() - t0) / steps

graphs = []
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for v in range(3):
        st.viewpoint_stack = [st.cameras[v]]
        tr.step(slot, verify_async=False)      # warm-up on the capture stream
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
for v in range(3):
    g = torch.cuda.CUDAGraph()
    st.viewpoint_stack = [st.cameras[v]]
    with torch.cuda.graph(g):
        tr.step(slot, verify_async=False)
    graphs.append(g)
for i in range(30):
    graphs[order[i % 3]].replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    graphs[order[i % 3]].replay()
torch.cuda.synchronize(); t_graph = (time.perf_counter() - t0) / steps
print("one-call step, eager launches : %.1f us/step (%.0f it/s)" % (1e6 * t_eager, 1 / t_eager))
print("one-call step, hipGraph replay: %.1f us/step (%.0f it/s)" % (1e6 * t_graph, 1 / t_graph))
BinningPolicy.reset("exact")
