#!/bin/bash
# round-3 GPU call: new tests, the matrix-pipe micro-benchmark, default bench line, host timeline of the drop-in loop with
# both bindings, the 2-rank path on the shared GPU
cd "$GRAFT_REPO_ROOT"
tools/ubench/mfma_mix > gpurun_out/r03_ubench_mfma_mix.txt 2>&1; grep -E "^   ->" gpurun_out/r03_ubench_mfma_mix.txt
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_edge_gpu.py tests/test_raster_gpu.py -x -q -s 2>&1 | grep -v "^$" | tail -8
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 400 gpurun_out/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print("value %.1f blocks %d | run-ahead %.1f | drop-in %.1f | 1k: %s | bwd %.4f fwd %.4f" % (d["value"], d["timed_blocks"], d["iters_per_sec_run_ahead"],
      d["iters_per_sec_dropin_reference_loop"], {k: round(v["iters_per_sec"], 1) for k, v in (d["iters_per_sec_1k"] or {}).items()},
      d["roofline"]["avg_kernel_ms"], d["roofline"]["composite_fwd"]["avg_kernel_ms"]))
PY
MI355GS_BINDING=compiled timeout 300 python tools/host_timeline.py 600 > gpurun_out/r03_dropin_host_timeline_compiled.txt 2>&1; cat gpurun_out/r03_dropin_host_timeline_compiled.txt
MI355GS_BINDING=ctypes timeout 300 python tools/host_timeline.py 600 > gpurun_out/r03_dropin_host_timeline_ctypes.txt 2>&1; head -3 gpurun_out/r03_dropin_host_timeline_ctypes.txt
timeout 300 python bench.py --gpus 2 --steps 100 --cpu-iters 0 > gpurun_out/bench_gpus2.json 2> gpurun_out/bench_gpus2.err; tail -c 300 gpurun_out/bench_gpus2.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_gpus2.json").read().strip().splitlines()[-1])
print("gpus2 value %.1f" % d["value"], json.dumps(d["multi_gpu"])[:900])
PY
