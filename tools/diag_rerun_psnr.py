"""How reproducible is a 200-iteration training run on the device (float atomics order differs from run to run)?  Trains the
scene of tests/test_scene_io_gpu.py several times from memory and from disk and prints the PSNRs."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests import test_scene_io_gpu as T
from instantsplat_amd.train import training

dev = torch.device("cuda:0")
with tempfile.TemporaryDirectory() as td:
    data = T._export(td, dev)
    for iters in (50, 100, 200):
        mem = [training(T._in_memory_scene(data, dev), dev, iterations=iters)["psnr_after"] for _ in range(4)]
        disk = [training(td, dev, iterations=iters, n_views=3)["psnr_after"] for _ in range(3)]
        print("iters %d  in-memory %s   from-disk %s" % (iters, " ".join("%.3f" % x for x in mem), " ".join("%.3f" % x for x in disk)), flush=True)
