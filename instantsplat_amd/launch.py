"""Scene-per-GPU launcher.

The reference's "multi-GPU" is a bash loop that polls nvidia-smi and starts one OS process per scene
(reference scripts/run_infer.sh:22-27,104-117); there is no communication on the train path.  Here:
one rank per GPU (torchrun / torch.distributed.run), rank i trains scene i, and ONE all_reduce
(RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests) combines the final metrics.
The message is 40 bytes, so the collective is latency-bound and link bandwidth is irrelevant.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m instantsplat_amd.launch --iterations 1000
"""
from __future__ import annotations

import argparse
import json
import os

import torch
import torch.distributed as dist


def pin_rank_to_cpu_slice(local_rank: int, local_world: int) -> list:
    """One process per GPU means N Python hosts on one socket: give each rank its own slice of the CPUs this job may use
    (sched_setaffinity) and cap its thread pools to it, so that eight eager launch loops do not migrate over and preempt
    each other — the scaling risk SURVEY.md 8e names is host contention, not the fabric.  Returns the CPUs kept."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:   # not Linux
        return []
    if local_world <= 1 or len(cpus) < local_world:
        return cpus
    per = len(cpus) // local_world
    mine = cpus[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    n_threads = max(1, min(per, 8))
    os.environ["OMP_NUM_THREADS"] = str(n_threads)
    torch.set_num_threads(n_threads)
    return mine


def gather_rank_reports(report: dict) -> list:
    """all_gather of one small dict per rank (rank 0 prints them); a list of one without a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, report)
        return out
    return [report]


def device_identity(dev) -> dict:
    """What distinguishes this rank's GPU from its neighbours' (index, PCI address / uuid when the runtime exposes them)."""
    if torch.device(dev).type != "cuda":
        return {"device": str(dev)}
    props = torch.cuda.get_device_properties(dev)
    ident = {"device": str(dev), "name": props.name}
    for attr in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
        v = getattr(props, attr, None)
        if v is not None:
            ident[attr] = str(v)
    return ident


def assert_one_rank_per_device(reports: list, device_count: int):
    """With at least as many GPUs as ranks, two ranks on one device is a launcher bug that would silently halve the
    measured scaling: refuse it."""
    # What decides it is the device index each rank selected on its host; uuid / PCI ids are reported, not trusted for this
    # (a runtime that fills them with the same value for every GPU must not turn a correct launch into an error).
    seen = [(r.get("host"), r["gpu"]["device"]) for r in reports]
    if device_count >= len(reports) and len(set(seen)) != len(seen):
        raise RuntimeError(f"ranks share a GPU although {device_count} are visible: {seen}")


def reduce_scene_metrics(psnr: float, n_images: int, iterations: int, seconds: float, device) -> dict:
    """SUM-reduce [psnr*n_images, n_images, iterations, 1] and MAX-reduce [seconds] over all ranks."""
    s = torch.tensor([psnr * n_images, float(n_images), float(iterations), 1.0], dtype=torch.float64, device=device)
    mx = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    s, mx = s.cpu(), mx.cpu()
    return dict(scenes=int(s[3]), mean_psnr=float(s[0] / s[1]), iterations=int(s[2]), max_seconds=float(mx[0]),
                aggregate_iters_per_sec=float(s[2] / mx[0]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=1000)
    ap.add_argument("--pointmap", type=int, default=256)
    ap.add_argument("--res", type=int, default=512)
    args = ap.parse_args()
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cpus = pin_rank_to_cpu_slice(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    reports = gather_rank_reports({"rank": rank, "gpu": device_identity(dev), "cpus": len(cpus)})
    assert_one_rank_per_device(reports, torch.cuda.device_count())
    from .synthetic import syn_pointmap
    from .train import training
    scene = syn_pointmap(3, args.pointmap, args.pointmap, args.res, args.res, seed=rank)  # scene i -> rank i
    if world > 1:
        dist.barrier()
    r = training(scene, dev, iterations=args.iterations)
    m = reduce_scene_metrics(r["psnr_after"], len(scene.cameras), args.iterations, r["seconds"], dev)
    if rank == 0:
        print(json.dumps(m))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
