"""Which NUMA node does each GPU hang off, what does the CPU topology look like in sysfs, and what would the launcher's CPU plan be
for 8 ranks on this box (GPUs 0-3 taken to be on node 0, 4-7 on node 1: only one GPU is visible here)?"""
import glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantsplat_amd import launch
for d in sorted(glob.glob("/sys/class/drm/card*/device")):
    try:
        real = os.path.basename(os.path.realpath(d))
        rd = lambda f: open(os.path.join(d, f)).read().strip()
        print(real, "numa_node", rd("numa_node"), "local_cpulist", rd("local_cpulist"))
    except OSError:
        pass
nodes = {}
for n in sorted(glob.glob("/sys/devices/system/node/node*")):
    nodes[int(os.path.basename(n)[4:])] = launch.parse_cpulist(open(os.path.join(n, "cpulist")).read())
    print(os.path.basename(n), open(os.path.join(n, "cpulist")).read().strip())
for c in list(range(0, 18)) + [31, 32, 63, 64, 127, 128, 129, 255]:
    print("cpu", c, "core", launch._physical_core_of(c), "L3", launch._l3_domain_of(c))
for i in range(torch.cuda.device_count()):
    print("torch device", i, "local cpus", launch.gpu_local_cpus(i)[:4], "...")
allowed = sorted(os.sched_getaffinity(0))
for world, near in ((8, lambda r: nodes[0 if r < 4 else 1]), (2, lambda r: nodes[0]), (1, lambda r: nodes[1])):
    plans = [launch.rank_cpu_plan(r, world, allowed, near) for r in range(world)]
    comp = [launch.compact_cpus(p) for p in plans]
    flat = [c for p in plans for c in p]
    print("world", world, "disjoint", len(flat) == len(set(flat)), "sizes", [len(p) for p in plans])
    for r in range(world):
        print("  rank", r, "slice", plans[r][:3], "..", plans[r][-1], "compact", comp[r], "L3 domains touched by the slice",
              len({launch._l3_domain_of(c) for c in plans[r]}))
