#!/bin/bash
# final-tree records of round 4 (second session): GPU tier, smoke, bench lines (driver args, default), the N > 1 timing path
# with real process groups (RCCL at world size 1, two ranks sharing the one GPU over gloo)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4g; python tools/box_probe.py 2>&1 | grep -E "device copy|v_fma_f32 +waves/SIMD 4" | tee gpurun_out/r4g/box.txt
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r4g/gpu_tier.log; cat gpurun_out/r4g/gpu_tier.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4g/bench_driver_args.json 2> gpurun_out/r4g/bench_driver_args.err; echo "bench1 rc $?"
timeout 900 python bench.py > gpurun_out/r4g/bench_default.json 2> gpurun_out/r4g/bench_default.err; echo "bench2 rc $?"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 --force-collectives --no-long-run --cpu-iters 0 > gpurun_out/r4g/bench_rccl_world1.json 2> gpurun_out/r4g/bench_rccl_world1.err; echo "bench3 rc $?"
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r4g/bench_gpus2_shared.json 2> gpurun_out/r4g/bench_gpus2_shared.err; echo "bench4 rc $?"
python - <<'PY'
import json
for f in ("bench_driver_args","bench_default","bench_rccl_world1","bench_gpus2_shared"):
    try:
        d=json.load(open("gpurun_out/r4g/%s.json" % f))
    except Exception as e:
        print(f, "unreadable", e); continue
    mg = d.get("multi_gpu") or {}
    print(f, d["n_gpus"], round(d["value"]), {k:round(v["iters_per_sec"]) for k,v in d["loops"].items()}, round(d["roofline"]["frac"],4), round(d["roofline"]["avg_kernel_ms"]*1e3,1),
          (d.get("cpu_baseline") or {}).get("value"), mg.get("rccl_version"), mg.get("solo_rank0_iters_per_sec"), mg.get("scaling_efficiency_vs_solo_rank0"),
          [round(r["iters_per_sec_median_block_own_clock"]) for r in mg.get("per_rank", [])])
PY
