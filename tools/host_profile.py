"""Ad-hoc: cProfile of the op-by-op autograd path (what an unmodified train.py gets), to see where the host time goes."""
import sys, time, cProfile, pstats, torch
sys.path.insert(0, '.')
from instantsplat_amd.arguments import OptimizationParams
from instantsplat_amd.synthetic import syn_pointmap
from instantsplat_amd.train import setup_training, train_iteration
dev = torch.device('cuda:0')
st = setup_training(syn_pointmap(3, 256, 256, 512, 512, seed=0), dev, opt=OptimizationParams(iterations=5000, pp_optimizer=True, optim_pose=True))
for _ in range(20): train_iteration(st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): train_iteration(st)
torch.cuda.synchronize(); print("ms/it", 1e3 * (time.perf_counter() - t0) / 200)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): train_iteration(st)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
