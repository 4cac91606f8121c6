"""GPU tier: the N > 1 line's collectives, proven on the one GPU a test box has.

`north_star` shards scenes one per GPU and uses RCCL only for the final metric reduction.  No 8-GPU node is reachable from
the build sandbox, but the init path is the same at world size 1: `init_process_group("nccl", device_id=...)` creates an RCCL
communicator on the MI355X, and barrier / SUM + MAX all_reduce / all_gather_object run through it.  These tests launch the
exact commands the driver uses (`python -m torch.distributed.run --nproc-per-node 1 ...`) with `--force-collectives`."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + args
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_through_rccl_at_world_size_one(gpu):
    out = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collectives", "--steps", "5", "--warmup", "2", "--pointmap", "64",
                     "--res", "128", "--cpu-iters", "0", "--no-long-run"])
    assert out["n_gpus"] == 1 and out["collective_backend"] == "nccl" and out["value"] > 0
    m = out["multi_gpu"]
    assert m["backend"] == "nccl" and m["rccl_version"], m          # an RCCL communicator existed and says which RCCL it is
    st = m["collective_selftest"]
    assert st["backend"] == "nccl" and st["world_size"] == 1 and st["rccl_version"] == m["rccl_version"]
    assert st["checked"] == ["barrier", "all_reduce SUM", "all_reduce MAX", "all_gather_object"] and st["all_reduce_40B_us"] > 0
    assert m["ranks_seen"] == m["world_size"] == 1 and m["per_rank"][0]["gpu"]["device"] == "cuda:0" and m["per_rank"][0]["host"]
    gpu_dir = os.path.join(ROOT, "gpurun_out")   # kept as a record when run through gpurun (copied to profiles/ by the builder)
    if os.path.isdir(gpu_dir):
        with open(os.path.join(gpu_dir, "rccl_world1_bench_line.json"), "w") as f:
            json.dump(out, f)


def test_scene_launcher_through_rccl_at_world_size_one(gpu):
    """`python -m instantsplat_amd.launch` — what replaces reference scripts/run_infer.sh:22-27,104-124 — under the same launcher."""
    out = _torchrun(["-m", "instantsplat_amd.launch", "--iterations", "30", "--pointmap", "48", "--res", "96", "--force-collectives"])
    assert out["scenes"] == 1 and out["iterations"] == 30 and out["aggregate_iters_per_sec"] > 0
    assert out["collectives"]["backend"] == "nccl" and out["collectives"]["rccl_version"]


def test_operators_on_a_side_stream_of_the_current_device(gpu):
    """One-GPU companion of test_operators_follow_their_tensors_device: cuda:0 current, tensors on cuda:0, the call made under a
    NON-default stream.  The library launches on the stream the binding hands it (`torch.cuda.current_stream(device)`), so the
    results must be complete after synchronising THAT stream only, and `_lib.on_device` must not switch devices."""
    from instantsplat_amd import _lib
    from tests.util import assert_raster_parity, run_blob_case
    dev = torch.device("cuda:0")
    assert _lib.on_device(dev) is _lib._NO_GUARD                      # same device: no guard object
    side = torch.cuda.Stream(device=dev)
    spin = torch.empty(64 << 20, device=dev)
    with torch.cuda.stream(side):
        assert _lib.stream_ptr(dev) == side.cuda_stream != torch.cuda.default_stream(dev).cuda_stream
        for _ in range(8):
            spin.normal_()                                            # the side stream is busy when the operators are enqueued
        out = run_blob_case(dev, 3000, 128, 96, 1, scale_mean=0.1)
        side.synchronize()
    assert_raster_parity(out)
    assert torch.cuda.current_device() == 0
