import torch, time, sys
sys.path.insert(0, '/root/repo')
from instantsplat_amd.simple_knn._C import distCUDA2
for n in (196608, 995328):
    p = torch.rand(n, 3, device='cuda')
    for _ in range(3): d = distCUDA2(p)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): d = distCUDA2(p)
    torch.cuda.synchronize(); print(n, "distCUDA2 ms", 1e3 * (time.perf_counter() - t0) / 20)
