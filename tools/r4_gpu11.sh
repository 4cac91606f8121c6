#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_raster_gpu.py tests/test_zz_reference_functions_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run > gpurun_out/r4/bench_v6.json 2> gpurun_out/r4/bench_v6.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4/bench_v6.json"))
for k,v in d["loops"].items(): print(k, round(v["iters_per_sec"],1), round(v["ms_per_step"]*1e3,1),"us")
PY
bash tools/prof.sh r4_loop_dropin python tools/loop_kernels.py dropin 400 > /dev/null
python - gpurun_out/r4_loop_dropin_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    c = int(r["Calls"])
    if c >= 100: print("  %6.2f / iter  avg %8.2f us  %s" % (c / 400.0, float(r["AverageNs"]) / 1e3, r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:100]))
PY
