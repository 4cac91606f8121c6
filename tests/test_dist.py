"""CPU tier, world_size 2 over gloo: the only collective on the path is the final metric reduction
 — sum of per-scene PSNR / iteration counts, max of per-rank seconds."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from instantsplat_amd.launch import reduce_scene_metrics


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # each rank "trained" its own scene: different PSNR, iteration count and wall time
    m = reduce_scene_metrics(psnr=30.0 + rank, n_images=3, iterations=100 * (rank + 1), seconds=2.0 + rank, device="cpu")
    if rank == 0:
        torch.save(m, out)
    dist.destroy_process_group()


def test_scene_metric_reduction_world2(tmp_path):
    out = str(tmp_path / "m.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    m = torch.load(out)
    assert m["scenes"] == 2
    assert abs(m["mean_psnr"] - 30.5) < 1e-9
    assert m["iterations"] == 300
    assert abs(m["max_seconds"] - 3.0) < 1e-9
    assert abs(m["aggregate_iters_per_sec"] - 300 / 3.0) < 1e-9


def test_scene_metric_reduction_single_process():
    m = reduce_scene_metrics(psnr=31.0, n_images=3, iterations=50, seconds=0.5, device="cpu")
    assert m["scenes"] == 1 and m["mean_psnr"] == 31.0 and m["aggregate_iters_per_sec"] == 100.0
