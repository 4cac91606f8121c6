#!/bin/bash
# round 4, GPU call 1: EXEC-half issue micro-benchmark, RCCL at world size 1, this box's baseline bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
timeout 120 tools/ubench/exec_half > gpurun_out/r4/ubench_exec_half.txt 2>&1; echo "ubench rc $?"
timeout 900 python -m pytest tests/test_rccl_gpu.py -x -q -m gpu > gpurun_out/r4/rccl_tests.log 2>&1; echo "rccl tests rc $?"
tail -5 gpurun_out/r4/rccl_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_baseline.json 2> gpurun_out/r4/bench_baseline.err; echo "bench rc $?"
cat gpurun_out/r4/ubench_exec_half.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4/bench_baseline.json"))
print({k:d[k] for k in ("value","ms_per_step","iters_per_sec_run_ahead","iters_per_sec_dropin_reference_loop")})
print(d["roofline"]["avg_kernel_ms"], d["roofline"]["composite_fwd"]["avg_kernel_ms"])
PY
