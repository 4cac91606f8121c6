#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tier_v4.log 2>&1; echo "gpu tier rc $?"; tail -3 gpurun_out/r4/gpu_tier_v4.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4/bench_v7_driver.json 2> gpurun_out/r4/bench_v7_driver.err; echo "bench rc $?"
timeout 600 python bench.py > gpurun_out/r4/bench_v7_default.json 2> gpurun_out/r4/bench_v7_default.err; echo "bench default rc $?"
python - <<'PY'
import json
for f in ("gpurun_out/r4/bench_v7_driver.json","gpurun_out/r4/bench_v7_default.json"):
    d=json.load(open(f))
    print(f, round(d["value"],1), round(d["ms_per_step"]*1e3,1), {k:round(v["iters_per_sec"]) for k,v in d["loops"].items()}, {k:round(v["iters_per_sec"]) for k,v in d["iters_per_sec_1k"].items()}, round(d["fps_reference_method"]["fps"]), round(d["cpu_baseline"]["value"],2))
    r=d["roofline"]; print("   ", round(r["avg_kernel_ms"]*1e3,1), round(r["frac"],4), round(r["frac_issue"],3), round(r["composite_fwd"]["avg_kernel_ms"]*1e3,1), round(r["composite_fwd"]["frac"],4), round(r["composite_fwd"]["compute"]["issue_frac_at_2.4GHz"],3))
PY
bash tools/prof.sh r4_c3 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run > /dev/null
bash tools/prof.sh r4_c4 python tools/c4_probe.py > /dev/null
timeout 600 python tools/configs.py > gpurun_out/r4/baseline_configs.txt 2>&1; tail -4 gpurun_out/r4/baseline_configs.txt | cut -c1-250
