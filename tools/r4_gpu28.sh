#!/bin/bash
# round 4, records of the final tree for profiles/
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4f
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r4f/gpu_tier.log; cat gpurun_out/r4f/gpu_tier.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4f/bench_driver_args.json 2> gpurun_out/r4f/bench_driver_args.err; echo "bench1 rc $?"
timeout 900 python bench.py > gpurun_out/r4f/bench_default.json 2> gpurun_out/r4f/bench_default.err; echo "bench2 rc $?"
bash tools/prof.sh r4f_c3 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run > /dev/null; cp gpurun_out/r4f_c3_kernel_stats.csv gpurun_out/r4f/
bash tools/prof.sh r4f_c4 python tools/c4_probe.py > /dev/null; cp gpurun_out/r4f_c4_kernel_stats.csv gpurun_out/r4f/
bash tools/prof.sh r4f_lk python tools/loop_kernels.py dropin 400 > /dev/null
python - <<'PY' > gpurun_out/r4f/dropin_loop_kernels.txt
import csv
for r in csv.DictReader(open("gpurun_out/r4f_lk_kernel_stats.csv")):
    c=int(r["Calls"])
    if c>=150: print("%5.2f / iter  avg %8.2f us  %s" % (c/400.0, float(r["AverageNs"])/1e3, r["Name"].replace("(anonymous namespace)::","")[:160]))
PY
timeout 900 python tools/configs.py > gpurun_out/r4f/configs.txt 2>&1; tail -5 gpurun_out/r4f/configs.txt
MI355GS_BINDING=compiled timeout 300 python tools/host_timeline.py 600 > gpurun_out/r4f/host_timeline.txt 2>&1
python - <<'PY'
import json
for f in ("bench_driver_args","bench_default"):
    d=json.load(open("gpurun_out/r4f/%s.json" % f))
    print(f, round(d["value"]), {k:round(v["iters_per_sec"]) for k,v in d["loops"].items()}, round(d["roofline"]["frac"],4), round(d["roofline"]["avg_kernel_ms"]*1e3,1), round(d["roofline"]["composite_fwd"]["avg_kernel_ms"]*1e3,1), d["cpu_baseline"]["value"], d.get("psnr_after_mean"), d.get("iters_per_sec_1k"))
PY
