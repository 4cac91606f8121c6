"""Which aten ops / kernels does ONE iteration of the drop-in loop run, and from where?  torch.profiler over N iterations of
the loop on the C3 scene: prints every device kernel with the CPU op that launched it.   Measurement helper, not product code."""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torch.profiler import profile, ProfilerActivity
from instantsplat_amd.arguments import OptimizationParams
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import setup_training, train_iteration

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
st = setup_training(syn_pointmap(3, 256, 256, 512, 512, seed=0), dev, opt=OptimizationParams(iterations=10 ** 9, pp_optimizer=True, optim_pose=True))
st.gaussians.oneupSHdegree = lambda: None
for _ in range(30):
    train_iteration(st)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(N):
        train_iteration(st)
    torch.cuda.synchronize()
ev = prof.events()
# device-side launches by name, and the CPU ops that own kernels
kern = collections.Counter()
owner = collections.defaultdict(collections.Counter)
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        top = e
        while top.cpu_parent is not None and top.cpu_parent.name not in ("ProfilerStep*",):
            top = top.cpu_parent
        for k in e.kernels:
            kern[k.name[:90]] += 1
            owner[k.name[:90]][(e.name[:60], top.name[:60])] += 1
for name, c in kern.most_common():
    print("%5.2f/iter  %s" % (c / N, name))
    for (op, top), n in owner[name].most_common(4):
        print("            %5.2f  op %-50s  under %s" % (n / N, op, top))
