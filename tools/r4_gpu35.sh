#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4f
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r4f/gpu_tier.log; cat gpurun_out/r4f/gpu_tier.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4f/bench_driver_args.json 2> gpurun_out/r4f/bench_driver_args.err; echo "bench1 rc $?"
timeout 900 python bench.py > gpurun_out/r4f/bench_default.json 2> gpurun_out/r4f/bench_default.err; echo "bench2 rc $?"
python - <<'PY'
import json
for f in ("bench_driver_args","bench_default"):
    d=json.load(open("gpurun_out/r4f/%s.json" % f))
    print(f, round(d["value"]), {k:round(v["iters_per_sec"]) for k,v in d["loops"].items()}, round(d["roofline"]["frac"],4), round(d["roofline"]["avg_kernel_ms"]*1e3,1), round(d["cpu_baseline"]["value"],2), {k:round(v["iters_per_sec"]) for k,v in d["iters_per_sec_1k"].items()})
PY
