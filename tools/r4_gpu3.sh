#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
timeout 600 python tools/diag_rerun_psnr.py > gpurun_out/r4/rerun_psnr.txt 2>&1; echo "diag rc $?"; cat gpurun_out/r4/rerun_psnr.txt | tail -5
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "commit_gate or fused_synced or run_ahead or gate" > gpurun_out/r4/commit_gate_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r4/commit_gate_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_commit_gate.json 2> gpurun_out/r4/bench_commit_gate.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4/bench_commit_gate.json"))
print({k:d[k] for k in ("value","ms_per_step","iters_per_sec_run_ahead","iters_per_sec_dropin_reference_loop")})
PY
