#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r4/gpu_tier.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4/bench_driver_args.json 2> gpurun_out/r4/bench_driver_args.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4/bench_driver_args.json"))
print(round(d["value"]), {k:round(v["iters_per_sec"]) for k,v in d["loops"].items()}, d["roofline"]["frac"], d["roofline"]["avg_kernel_ms"], d["cpu_baseline"])
PY
