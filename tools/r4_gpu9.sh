#!/bin/bash
# round 4, GPU call 9: ABI v7 (visible from the kernel, accumulators cleared by the projection kernel, shared zero f_rest gradient)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tier_v3.log 2>&1; echo "gpu tier rc $?"; tail -3 gpurun_out/r4/gpu_tier_v3.log
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run > gpurun_out/r4/bench_v5.json 2> gpurun_out/r4/bench_v5.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4/bench_v5.json"))
for k,v in d["loops"].items(): print(k, round(v["iters_per_sec"],1), round(v["ms_per_step"]*1e3,1),"us")
PY
MI355GS_BINDING=compiled timeout 300 python tools/host_timeline.py 600 > gpurun_out/r4/host_timeline.txt 2>&1; tail -22 gpurun_out/r4/host_timeline.txt
bash tools/prof.sh r4_c3 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run > /dev/null
