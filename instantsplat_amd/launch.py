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
import socket

# (multi-process GPU work on this pool needs dmabuf IPC; the boxes export it — this only covers a shell that lost it, and must
# precede whatever initialises the HSA runtime)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def _physical_core_of(cpu: int):
    """(package, core) of a logical CPU from sysfs, or None where the kernel does not say."""
    base = f"/sys/devices/system/cpu/cpu{cpu}/topology/"
    try:
        with open(base + "physical_package_id") as f:
            pkg = int(f.read())
        with open(base + "core_id") as f:
            return pkg, int(f.read())
    except (OSError, ValueError):
        return None


def cpu_slices(cpus: list, n: int, core_of=_physical_core_of) -> list:
    """`cpus` cut into n disjoint slices of whole PHYSICAL cores, neighbours in (package, core) order.  Linux numbers the
    second hardware thread of every core after all the first ones (0..C-1, then C..2C-1), so equal runs of the sorted ids
    would hand rank r and rank r + n/2 the two threads of the SAME cores — two spinning launch loops sharing an execution
    unit.  Falls back to equal runs of ids when the topology is not readable or has fewer cores than slices."""
    cpus = sorted(cpus)
    cores = {}
    for c in cpus:
        key = core_of(c)
        if key is None:
            cores = None
            break
        cores.setdefault(key, []).append(c)
    if cores is None or len(cores) < n:
        per = len(cpus) // n
        return [cpus[r * per:(r + 1) * per] for r in range(n)]
    order = sorted(cores)
    per = len(order) // n
    return [sorted(c for key in order[r * per:(r + 1) * per] for c in cores[key]) for r in range(n)]


def parse_cpulist(text: str) -> list:
    """'0-63,128-191' -> [0, .., 63, 128, .., 191] (the kernel's cpulist format)"""
    out = []
    for part in text.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            out.extend(range(int(lo), int(hi or lo) + 1))
    return sorted(set(out))


def gpu_local_cpus(device_index: int):
    """The CPUs of the NUMA node GPU `device_index` hangs off (sysfs `local_cpulist` of its PCI function), or None where
    that cannot be read.  Measured on an MI355X box (2 sockets, the GPU on node 1; tools/ab_numa.sh,
    profiles/r04_ab_numa_affinity.txt): the host-bound loops — train.py's loss spelled out in eager PyTorch around the
    drop-in operators — run 12-14 % faster with the process kept on the GPU's node than left to the scheduler."""
    try:
        p = torch.cuda.get_device_properties(device_index)
        addr = "%04x:%02x:%02x.0" % (int(p.pci_domain_id), int(p.pci_bus_id), int(p.pci_device_id))
        with open(f"/sys/bus/pci/devices/{addr}/local_cpulist") as f:
            cpus = parse_cpulist(f.read())
        return cpus or None
    except Exception:   # no such attribute / file / a CPU device: the caller falls back to topology-blind slices
        return None


def rank_cpu_plan(local_rank: int, local_world: int, allowed: list, local_cpus_of_rank=None, core_of=_physical_core_of) -> list:
    """The CPUs rank `local_rank` of `local_world` ranks on this node keeps.  local_cpus_of_rank(r) -> the CPUs next to rank r's
    GPU (or None): ranks whose GPUs share a NUMA node split THAT node's physical cores among themselves, in rank order; without
    the information (or when a node's CPUs are not in `allowed`) all ranks split `allowed` into equal runs of physical cores."""
    allowed = sorted(allowed)
    if local_world <= 0 or len(allowed) < max(local_world, 1):
        return allowed
    if local_cpus_of_rank is not None:
        near = [local_cpus_of_rank(r) for r in range(local_world)]
        if all(n is not None for n in near):
            keep = set(allowed)
            near = [tuple(c for c in n if c in keep) for n in near]
            mates = [r for r in range(local_world) if near[r] == near[local_rank]]
            if all(len(n) >= local_world for n in near):   # every rank decides from the same table, so the slices are disjoint
                return cpu_slices(list(near[local_rank]), len(mates), core_of=core_of)[mates.index(local_rank)]
    if local_world == 1:
        return allowed
    return cpu_slices(allowed, local_world, core_of=core_of)[local_rank]


def _l3_domain_of(cpu: int):
    """The CPUs that share cpu's last-level cache (sysfs index3/shared_cpu_list: a CCD on an EPYC), as a tuple, or None."""
    try:
        with open(f"/sys/devices/system/cpu/cpu{cpu}/cache/index3/shared_cpu_list") as f:
            return tuple(parse_cpulist(f.read()))
    except (OSError, ValueError):
        return None


def compact_cpus(cpus: list, core_of=_physical_core_of, l3_of=_l3_domain_of, min_cores: int = 4) -> list:
    """One hardware thread per physical core of ONE last-level-cache domain of `cpus` (the first with at least `min_cores`
    cores).  The threads that matter to a launch loop — the Python thread, the autograd engine's device thread, the HIP
    runtime's helpers — hand work to each other every few microseconds; on one CCD they meet in a shared L3, and each has a
    core to itself.  Measured (tools/ab_numa2.sh, profiles/r04_ab_numa_affinity.txt): the drop-in loop + 5 %, train.py's loss in
    eager PyTorch + 10 % over the whole NUMA node.  Returns `cpus` unchanged where the cache topology is not readable."""
    domains = {}
    for c in sorted(cpus):
        dom, core = l3_of(c), core_of(c)
        if dom is None or core is None:
            return sorted(cpus)
        domains.setdefault(dom, {}).setdefault(core, c)   # first (lowest) hardware thread of each core
    for dom in sorted(domains):
        if len(domains[dom]) >= min_cores:
            return sorted(domains[dom].values())
    return sorted(cpus)


def pin_mode() -> str:
    """MI355GS_PIN = "node" (default) | "compact" | "off": how far the launcher / bench narrow a rank's CPUs.
    node: whole physical cores of the NUMA node of the rank's GPU (the ranks of that node split its cores).
    compact: on top of that ONE last-level-cache domain, a core per thread — the fastest host loop on a quiet box (the drop-in
        loop + 5 % over "node"), but a hard pin to eight particular CPUs cannot dodge another tenant's load on them the way the
        scheduler does for a wider mask: on a shared box a run pinned like this once took more than ten times its usual time.
        For a node that is the job's own.
    off: leave the affinity alone."""
    m = os.environ.get("MI355GS_PIN", "node").strip().lower()
    return m if m in ("node", "compact", "off") else "node"


def pin_rank_to_cpu_slice(local_rank: int, local_world: int, device_of_rank=None, compact: bool = False) -> list:
    """One process per GPU means N Python hosts on one box: give each rank its own slice of the CPUs this job may use
    (sched_setaffinity) — whole physical cores, on the NUMA node of its GPU when `device_of_rank(r)` (rank -> device index)
    is given and sysfs knows the node — and cap its thread pools to it, so that eight eager launch loops do not migrate over
    and preempt each other: the scaling risk SURVEY.md 8e names is host contention, not the fabric.  A single rank with a
    device is kept on its GPU's node.  compact: narrow the slice further to one last-level-cache domain, one hardware thread
    per core (`compact_cpus`).  Returns the CPUs kept."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:   # not Linux
        return []
    near = (lambda r: gpu_local_cpus(device_of_rank(r))) if device_of_rank is not None else None
    if local_world <= 0 or len(cpus) < local_world or (local_world == 1 and near is None):
        return cpus            # (0: a multi-node job without LOCAL_WORLD_SIZE — not sliced)
    mine = rank_cpu_plan(local_rank, local_world, cpus, near)
    if compact and mine:
        mine = compact_cpus(mine)
    if not mine or mine == cpus:
        return cpus
    os.sched_setaffinity(0, mine)
    if local_world > 1:   # (a single rank keeps its thread pools: it has a whole NUMA node to itself)
        n_threads = max(1, min(len(mine), 8))
        os.environ["OMP_NUM_THREADS"] = str(n_threads)
        torch.set_num_threads(n_threads)
    return mine


def local_world_size(world: int) -> int:
    """Ranks on THIS node.  torchrun exports LOCAL_WORLD_SIZE; without it a one-node job has `world` of them, and a job that
    spans nodes (RANK >= what one node can hold is unknowable here) is not sliced at all: 0 = do not pin."""
    v = os.environ.get("LOCAL_WORLD_SIZE")
    if v is not None:
        return int(v)
    return world if int(os.environ.get("GROUP_WORLD_SIZE", os.environ.get("NNODES", "1")) or 1) <= 1 else 0


def init_collectives(backend: str, rank: int, world: int, dev) -> None:
    """The one process group of the path (backend "nccl" = RCCL over xGMI on ROCm).  Also valid at world size 1: the
    communicator is created, and the collectives below run through RCCL on the one device — which is how the init path
    is proven on a 1-GPU box before the 8-GPU run (`--force-collectives`)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:   # no launcher (single process): any free port will do
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)


def collective_selftest(dev, rounds: int = 50) -> dict:
    """Every collective the path uses, once, with a known answer — barrier, SUM and MAX all_reduce of a float64 vector,
    all_gather_object — plus the latency of the 40-byte all_reduce the final metric reduction is (SURVEY.md 8e:
    latency-bound).  Raises if an answer is wrong; returns what was measured."""
    import time
    world, rank = dist.get_world_size(), dist.get_rank()
    backend = dist.get_backend()
    red_dev = torch.device(dev) if backend == "nccl" else torch.device("cpu")
    sync = (lambda: torch.cuda.synchronize(red_dev)) if red_dev.type == "cuda" else (lambda: None)
    dist.barrier()
    s = torch.tensor([float(rank + 1), 1.0, 2.5, 0.0, 0.0], dtype=torch.float64, device=red_dev)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    mx = torch.tensor([float(rank)], dtype=torch.float64, device=red_dev)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    got = [None] * world
    dist.all_gather_object(got, {"rank": rank})
    assert s.tolist() == [world * (world + 1) / 2.0, float(world), 2.5 * world, 0.0, 0.0], s.tolist()
    assert mx.item() == float(world - 1) and [g["rank"] for g in got] == list(range(world)), (mx.item(), got)
    sync()
    t0 = time.perf_counter()
    for _ in range(rounds):
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    sync()
    us = 1e6 * (time.perf_counter() - t0) / rounds
    rccl = ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None
    return {"backend": backend, "world_size": world, "rccl_version": rccl, "all_reduce_40B_us": us,
            "checked": ["barrier", "all_reduce SUM", "all_reduce MAX", "all_gather_object"]}


def gather_rank_reports(report: dict) -> list:
    """all_gather of one small dict per rank (rank 0 prints them); a list of one without a process group."""
    report.setdefault("host", socket.gethostname())
    if dist.is_available() and dist.is_initialized():
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
    if dist.is_available() and dist.is_initialized():
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
    ap.add_argument("--force-collectives", action="store_true",
                    help="create the RCCL process group and run the collectives even at world size 1 (init-path check on one GPU)")
    args = ap.parse_args()
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    n_dev = max(torch.cuda.device_count(), 1)
    cpus = (pin_rank_to_cpu_slice(local, local_world_size(world), device_of_rank=lambda r: r % n_dev, compact=pin_mode() == "compact")
            if pin_mode() != "off" else sorted(os.sched_getaffinity(0)))
    selftest = None
    if world > 1 or args.force_collectives:
        init_collectives("nccl", rank, world, dev)
        selftest = collective_selftest(dev)
    reports = gather_rank_reports({"rank": rank, "host": socket.gethostname(), "gpu": device_identity(dev), "cpus": len(cpus)})
    assert_one_rank_per_device(reports, torch.cuda.device_count())
    from .synthetic import syn_pointmap
    from .train import training
    scene = syn_pointmap(3, args.pointmap, args.pointmap, args.res, args.res, seed=rank)  # scene i -> rank i
    if dist.is_initialized():
        dist.barrier()
    r = training(scene, dev, iterations=args.iterations)
    m = reduce_scene_metrics(r["psnr_after"], len(scene.cameras), args.iterations, r["seconds"], dev)
    m["collectives"] = selftest
    if rank == 0:
        print(json.dumps(m))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
