"""CPU tier, world_size 2 over gloo: the only collective on the path is the final metric reduction
 — sum of per-scene PSNR / iteration counts, max of per-rank seconds."""
import json
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from instantsplat_amd.launch import reduce_scene_metrics

ROOT_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line_and_full(stdout):
    """THE line (exactly one, nothing else on stdout, under 8 KB, parses, carries what the driver reads — VERDICT r5 #1: round 5's
    23.8 KB line came back from the driver as `parsed: null`) and the long-form record it points to."""
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and [l for l in stdout.splitlines() if l.strip()] == lines, stdout
    assert len(lines[0].encode()) < 8192, len(lines[0])
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in out, k
    assert out["config"]["workload"] and "model" not in out["config"] and "unpinned" in out["config"]["parity"]
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and "frac" in rf and "achieved" in rf and "traffic" in rf
    assert "cpu_baseline" in out and "value_without_host_tricks" in out
    assert all(isinstance(v, (int, float)) for v in out["loops"].values()) and len(out["loops"]) == 7     # one number per loop
    assert "line_trimmed" not in out
    with open(os.path.join(ROOT_, out["full_record"])) as fh:
        full = json.load(fh)
    assert full["value"] == full["loops"]["dropin_reference_loop_train_py_loss"]["iters_per_sec"]
    assert abs(full["value"] - out["value"]) <= 1e-4 * full["value"]
    return out, full


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


def test_bench_gpus2_spawns_two_ranks_end_to_end(emu_lib_path):
    """`python bench.py --gpus 2` with no launcher around it must start two ranks itself (torch.distributed.run), train one scene
    per rank, barrier, MAX-reduce the time, SUM-reduce PSNR and print ONE line with n_gpus = 2 (`--gpus` used to be
    parsed and ignored).  CPU tier: the ranks run the emulated kernels on a toy scene over gloo — the plumbing is what is
    tested; the same code path runs over RCCL on GPUs."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--pointmap", "6",
                        "--res", "32", "--cpu-iters", "0", "--emulated-kernels", emu_lib_path], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    # rank 0 only, and nothing else on stdout: gloo's connection announcements ("[Gloo] Rank 0 is connected to ...") go to stderr
    out, full = _line_and_full(r.stdout)
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["collective_backend"] == "gloo" and "EMULATED" in out["data"]
    assert out["value"] > 0 and abs(out["value"] - 2 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-3 * out["value"]   # (5 significant digits in the line)
    assert abs(full["value"] - 2 * 3 / (full["ms_per_step"] * 3e-3)) < 1e-6 * full["value"] and out["metric"] == "train_iters_per_sec"
    assert out["psnr_after_mean"] == out["psnr_after_mean"] and out["psnr_after_mean"] > 5.0
    assert out["config"]["parallelism"] == "scene-per-gpu x2"
    # what makes the 8-GPU run informative: who ran where, per-rank rates on the rank's own clock, the ranks'
    # CPU slices, and the line's own N = 1 reference (rank 0 alone on the box) for the scaling efficiency
    for m in (out["multi_gpu"], full["multi_gpu"]):   # the line's short table and the full record's
        assert m["ranks_seen"] == m["world_size"] == 2 and m["backend"] == "gloo"
        assert sorted(r["rank"] for r in m["per_rank"]) == [0, 1]
        assert all(r["iters_per_sec_median_block_own_clock"] > 0 and r["cpus"] >= 1 and "gpu" in r and r["psnr_after"] > 5.0 for r in m["per_rank"])
        if len(os.sched_getaffinity(0)) >= 2:   # each rank pinned itself to its own slice of the CPUs
            assert m["per_rank"][0]["first_cpu"] != m["per_rank"][1]["first_cpu"]
        assert m["solo_rank0_iters_per_sec"] > 0 and m["scaling_efficiency_vs_solo_rank0"] > 0
    assert out["multi_gpu"]["hosts"] == [socket.gethostname()]
    assert out["timed_blocks"] >= 1 and len(full["block_seconds"]) == full["timed_blocks"] and "read-back" in full["loop"]


def test_bench_gpus8_eight_ranks_report_kernel_times_and_disjoint_cpu_slices(emu_lib_path):
    """The driver's one shot at an 8-GPU node, rehearsed: `bench.py --gpus 8` starts eight ranks, and the line carries for EVERY rank
    its rate on its own clock AND its own composite-kernel event times with their roofline fractions and its box tag — the
    north-star asks for roofline evidence at 1/2/4/8 GPUs, not only rates — from disjoint CPU slices, within a
    wall time that fits the driver's budget many times over.  Emulated kernels over gloo: the plumbing is what is tested."""
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--pointmap", "6",
                        "--res", "32", "--cpu-iters", "0", "--emulated-kernels", emu_lib_path], env=env, capture_output=True, text=True,
                       timeout=1500)
    wall = time.time() - t0
    assert r.returncode == 0, r.stderr[-12000:]   # (the launcher's own summary is the last 3 KB: the ranks' messages are above it)
    out, full = _line_and_full(r.stdout)   # eight per-rank rows and still under 8 KB
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["config"]["parallelism"] == "scene-per-gpu x8"
    assert abs(full["value"] - 8 * 2 / (full["ms_per_step"] * 2e-3)) < 1e-6 * full["value"]
    m = out["multi_gpu"]
    assert m["ranks_seen"] == m["world_size"] == 8 and sorted(x["rank"] for x in m["per_rank"]) == list(range(8))
    for x in m["per_rank"]:
        assert x["iters_per_sec_median_block_own_clock"] > 0 and x["psnr_after"] > 5.0 and x["R_eff"] > 0
        for k in ("composite_bwd_avg_ms", "composite_fwd_avg_ms", "composite_bwd_frac_hbm", "composite_fwd_frac_hbm", "device_copy_TB_per_s"):
            assert k in x, k   # (emulated kernels are not timed by HIP events: the fields are there, their values are the GPU run's business)
    for x in full["multi_gpu"]["per_rank"]:
        for k in ("composite_fwd_render_only_avg_ms", "box", "host"):
            assert k in x, k
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 8:   # each rank pinned itself to its own slice of the CPUs
        firsts = [x["first_cpu"] for x in m["per_rank"]]
        assert len(set(firsts)) == 8, firsts
    assert m["solo_rank0_iters_per_sec"] > 0 and m["scaling_efficiency_vs_solo_rank0"] > 0
    assert wall < 900, wall   # the driver allows 1800 s for a run; on a GPU node a rank takes about a minute
    with open(os.path.join(root, "gpurun_out", "r06_bench_gpus8_emulated_dry_run.json") if os.path.isdir(os.path.join(root, "gpurun_out")) else os.devnull, "w") as fh:
        fh.write(json.dumps(out) + "\n")


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 4" in r.stderr


def test_rank_device_collision_check():
    """Two ranks on one device index of one host is refused when enough GPUs are visible; equal uuids alone (a runtime that
    reports the same identity for every GPU) and the same index on different hosts are not."""
    import pytest
    from instantsplat_amd.launch import assert_one_rank_per_device
    def rep(host, dev, uuid="u"):
        return {"host": host, "gpu": {"device": dev, "uuid": uuid}}
    assert_one_rank_per_device([rep("a", "cuda:0"), rep("a", "cuda:1")], 8)          # same uuid, different devices: fine
    assert_one_rank_per_device([rep("a", "cuda:0"), rep("b", "cuda:0")], 8)          # two hosts
    assert_one_rank_per_device([rep("a", "cuda:0"), rep("a", "cuda:0")], 1)          # a 1-GPU box sharing on purpose
    with pytest.raises(RuntimeError):
        assert_one_rank_per_device([rep("a", "cuda:0", "x"), rep("a", "cuda:0", "y")], 8)


def test_bench_force_collectives_at_world_size_one(emu_lib_path):
    """`--force-collectives` at N = 1: the process group is created and every collective of the N > 1 line runs through it (gloo
    here; the GPU tier runs the same command over RCCL, tests/test_rccl_gpu.py)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-collectives", "--steps", "3", "--warmup", "1",
                        "--pointmap", "6", "--res", "32", "--cpu-iters", "0", "--emulated-kernels", emu_lib_path], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out, full = _line_and_full(r.stdout)
    assert out["n_gpus"] == 1 and out["collective_backend"] == "gloo"
    m = out["multi_gpu"]
    assert m["ranks_seen"] == m["world_size"] == 1 and m["hosts"] and full["multi_gpu"]["per_rank"][0]["host"]
    assert m["collectives_checked"] == ["barrier", "all_reduce SUM", "all_reduce MAX", "all_gather_object"]
    assert "composite_bwd_frac_hbm" in m["per_rank"][0]
    assert out["attempts"] == 1 and out["attempt_seconds"] > 0   # a single-process run goes through the supervising parent


def test_bench_kills_a_stalled_attempt_and_tries_once_more(emu_lib_path):
    """A run that produces no line within --attempt-seconds is killed and repeated once (unpinned); when
    that stalls too the command fails instead of hanging — and leaves no process behind."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MI355GS_BENCH_CHILD")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--pointmap", "6", "--res", "32",
                        "--cpu-iters", "0", "--emulated-kernels", emu_lib_path, "--attempt-seconds", "0.5"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and r.stdout.strip() == ""
    assert r.stderr.count("was killed") == 2 and "attempt 2" in r.stderr
    assert "most recent call first" in r.stderr   # the stalled child said where it stood (faulthandler) before it was killed


def test_launch_report_carries_the_host_and_multi_node_is_not_sliced(monkeypatch):
    """`launch.main()`'s report had no "host", so two nodes' cuda:0 collided in assert_one_rank_per_device; and the CPU
    slices must come from LOCAL_WORLD_SIZE, not from the global world size, on a multi-node job."""
    from instantsplat_amd import launch
    rep = launch.gather_rank_reports({"rank": 0, "gpu": {"device": "cuda:0"}})
    assert rep[0]["host"] == socket.gethostname()
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.setenv("NNODES", "2")
    assert launch.local_world_size(8) == 0 and launch.pin_rank_to_cpu_slice(0, 0) == sorted(os.sched_getaffinity(0))
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    assert launch.local_world_size(8) == 4


def test_cpu_slices_are_whole_physical_cores():
    """Two sockets x 4 cores x 2 hardware threads in Linux's numbering (cpu c and c + 8 share a core): no two ranks may get
    threads of one core, every CPU is handed out once, and an unreadable topology degrades to equal runs of ids."""
    from instantsplat_amd.launch import cpu_slices
    core_of = lambda c: ((c % 8) // 4, c % 4)
    for n in (2, 4, 8):
        sl = cpu_slices(list(range(16)), n, core_of=core_of)
        assert sorted(c for s in sl for c in s) == list(range(16)) and len({len(s) for s in sl}) == 1
        owners = {}
        for r, s in enumerate(sl):
            for c in s:
                assert owners.setdefault(core_of(c), r) == r, (n, sl)
    assert cpu_slices(list(range(16)), 8, core_of=core_of)[0] == [0, 8]
    assert cpu_slices(list(range(16)), 4, core_of=lambda c: None) == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]
    assert cpu_slices(list(range(4)), 4, core_of=lambda c: (0, c // 2)) == [[0], [1], [2], [3]]   # fewer cores than slices
    assert [len(s) for s in cpu_slices(sorted(os.sched_getaffinity(0)), 1)] == [len(os.sched_getaffinity(0))]


def test_rank_cpu_plan_keeps_a_rank_on_its_gpus_numa_node():
    """Two NUMA nodes x 4 cores x 2 hardware threads (cpu c and c + 8 share a core; node = core // 4).  GPUs enumerated against
    the node order (ranks 0, 1 next to node 1; ranks 2, 3 next to node 0): every rank gets whole cores of ITS GPU's node, the
    slices are disjoint, a single rank keeps the whole node of its GPU, and missing information degrades to topology-blind slices."""
    from instantsplat_amd.launch import parse_cpulist, rank_cpu_plan
    assert parse_cpulist("0-3,8-11\n") == [0, 1, 2, 3, 8, 9, 10, 11] and parse_cpulist("5") == [5] and parse_cpulist("") == []
    core_of = lambda c: ((c % 8) // 4, c % 8)
    node_cpus = {0: parse_cpulist("0-3,8-11"), 1: parse_cpulist("4-7,12-15")}
    near = lambda r: node_cpus[1 if r < 2 else 0]
    plans = [rank_cpu_plan(r, 4, list(range(16)), near, core_of=core_of) for r in range(4)]
    assert plans == [[4, 5, 12, 13], [6, 7, 14, 15], [0, 1, 8, 9], [2, 3, 10, 11]]
    assert rank_cpu_plan(0, 1, list(range(16)), near, core_of=core_of) == node_cpus[1]          # N = 1: the GPU's node
    assert rank_cpu_plan(0, 2, list(range(16)), lambda r: node_cpus[1], core_of=core_of) == [4, 5, 12, 13]   # two ranks sharing one GPU
    assert rank_cpu_plan(1, 2, list(range(16)), lambda r: node_cpus[1], core_of=core_of) == [6, 7, 14, 15]
    blind = [rank_cpu_plan(r, 4, list(range(16)), lambda r: None, core_of=core_of) for r in range(4)]
    assert blind == [[0, 1, 8, 9], [2, 3, 10, 11], [4, 5, 12, 13], [6, 7, 14, 15]]
    assert rank_cpu_plan(0, 1, list(range(16)), None, core_of=core_of) == list(range(16))
    # the job may only use node 0's CPUs (a cgroup): a GPU on node 1 has nothing local to offer -> topology-blind slices of what is allowed
    assert rank_cpu_plan(0, 2, node_cpus[0], near, core_of=core_of) == [0, 1, 8, 9]


def test_compact_cpus_is_one_cache_domain_one_thread_per_core():
    """16 logical CPUs: 8 cores x 2 threads (c and c + 8), two L3 domains of 4 cores.  A slice of whole cores shrinks to the first
    domain's cores, first hardware thread of each; a domain with too few cores is passed over; unreadable topology changes nothing."""
    from instantsplat_amd.launch import compact_cpus
    core_of = lambda c: (0, c % 8)
    l3_of = lambda c: tuple(sorted([x for x in range(16) if (x % 8) // 4 == (c % 8) // 4]))
    assert compact_cpus(list(range(16)), core_of, l3_of) == [0, 1, 2, 3]
    assert compact_cpus([4, 5, 6, 7, 12, 13, 14, 15], core_of, l3_of) == [4, 5, 6, 7]
    assert compact_cpus([3, 11, 4, 5, 6, 7, 12, 13, 14, 15], core_of, l3_of) == [4, 5, 6, 7]     # one core of domain 0: too few
    assert compact_cpus([2, 3, 10, 11], core_of, l3_of, min_cores=2) == [2, 3]
    assert compact_cpus([0, 1, 2], core_of, lambda c: None) == [0, 1, 2]
    assert compact_cpus([0, 8], core_of, l3_of) == [0, 8]                                            # nothing qualifies: unchanged


def test_compact_line_never_exceeds_the_limit_and_says_what_it_shed(tmp_path):
    """The last-resort trimmer of bench.compact_line: a record whose optional blocks would push the line past 8 KB (sixty-four
    per-rank rows, a bloated small-kernel block) still yields ONE parsable line under the limit that keeps every contract key,
    `roofline.frac` and `cpu_baseline`, and flags `line_trimmed`; NaN / inf become null (JSON has neither)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT_, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    with open(os.path.join(ROOT_, "profiles", "r06_bench_default_full_record.json")) as fh:
        full = json.load(fh)
    line = bench.compact_line(full, "gpurun_out/x.json")
    assert len(line) < 4096 and "line_trimmed" not in json.loads(line)
    rank = {"rank": 0, "host": "h" * 40, "gpu": {"device": "cuda:0"}, "cpus": 16, "first_cpu": 0, "iters_per_sec_median_block_own_clock": 3512.123456,
            "psnr_after": 50.123456, "composite_bwd_avg_ms": 0.1071234, "composite_fwd_avg_ms": 0.0741234, "composite_bwd_frac_hbm": 0.1051234,
            "composite_fwd_frac_hbm": 0.0591234, "R_eff": 756123.4, "box": {"device_copy_TB_per_s": 5.41234}}
    full["multi_gpu"] = {"per_rank": [dict(rank, rank=i) for i in range(64)], "ranks_seen": 64, "world_size": 64, "backend": "nccl", "rccl_version": "2.26.6",
                         "collective_selftest": {"checked": ["barrier"]}, "solo_rank0_iters_per_sec": 3500.0, "scaling_efficiency_vs_solo_rank0": float("nan")}
    full["roofline"]["small_kernels"] = {"k_%d" % i: {"avg_kernel_ms": 0.01, "frac": 0.05, "traffic": 1.0e6} for i in range(40)}
    line = bench.compact_line(full, "gpurun_out/x.json")
    out = json.loads(line)
    assert len(line) < bench.LINE_LIMIT == 8192 and out["line_trimmed"] is True
    assert out["roofline"]["frac"] > 0 and out["cpu_baseline"]["value"] > 0 and out["value"] > 0 and out["config"]["workload"]
    assert out["multi_gpu"]["scaling_efficiency_vs_solo_rank0"] is None and "NaN" not in line
