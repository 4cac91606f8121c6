#!/bin/bash
# usage (on the GPU box): tools/perf.sh   -> short bench line digest only (no tests)
cd "$GRAFT_REPO_ROOT"
timeout 200 python bench.py --cpu-iters 0 --steps 150 --warmup 20 > gpurun_out/quick.json 2> gpurun_out/quick.err || tail -5 gpurun_out/quick.err
python - <<PY
import json
d = json.loads(open("gpurun_out/quick.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("it/s %.1f  ms/step %.4f  bwd %.4f ms  fwd %.4f ms psnr %.2f" % (d["value"], d["ms_per_step"], r["avg_kernel_ms"], r["composite_fwd"]["avg_kernel_ms"], d["psnr_after_mean"]))
PY
