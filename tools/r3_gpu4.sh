#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^$" | tail -6
timeout 600 python bench.py --cpu-iters 0 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 400 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print("value %.1f blocks %d | run-ahead %.1f | drop-in %.1f | 1k: %s | bwd %.4f fwd %.4f" % (d["value"], d["timed_blocks"], d["iters_per_sec_run_ahead"],
      d["iters_per_sec_dropin_reference_loop"], {k: round(v["iters_per_sec"], 1) for k, v in (d["iters_per_sec_1k"] or {}).items()},
      d["roofline"]["avg_kernel_ms"], d["roofline"]["composite_fwd"]["avg_kernel_ms"]))
PY
MI355GS_BINDING=compiled timeout 300 python tools/host_timeline.py 600 2>&1 | grep -v amdgpu
