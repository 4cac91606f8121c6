#!/bin/bash
# usage (on the GPU box): tools/quick.sh [pytest -k expr]   -> raster parity tests + a short bench line digest
cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_raster_gpu.py tests/test_edge_gpu.py tests/test_properties_gpu.py -x -q 2>&1 | tail -3
timeout 200 python bench.py --cpu-iters 0 > gpurun_out/quick.json 2> gpurun_out/quick.err || tail -5 gpurun_out/quick.err
python - <<PY
import json
d = json.loads(open("gpurun_out/quick.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("it/s %.1f  ms/step %.4f  bwd %.4f ms  sync-loop %.1f  autograd %.1f  raster %.4f ms  psnr %.2f" % (
    d["value"], d["ms_per_step"], r["avg_kernel_ms"], d["iters_per_sec_with_per_iteration_loss_readback"],
    d["iters_per_sec_autograd_path"], d["rasterize_ms_per_frame"], d["psnr_after_mean"]))
print("fwd %.4f ms" % r["composite_fwd"]["avg_kernel_ms"])
PY
