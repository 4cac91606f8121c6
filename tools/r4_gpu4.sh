#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4/bench_v2_driver.json 2> gpurun_out/r4/bench_v2_driver.err ) 2>&1 | grep real; echo "bench rc $?"
tail -3 gpurun_out/r4/bench_v2_driver.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4/bench_v2_driver.json"))
print({k:d[k] for k in ("value","ms_per_step","iters_per_sec_one_call_synced","iters_per_sec_run_ahead","timed_iterations","legs_skipped")})
for k,v in d["loops"].items(): print(k, round(v["iters_per_sec"],1), [round(x*1e3,2) for x in v["block_seconds"][::6]])
r=d["roofline"]; print({k:r[k] for k in ("bound","frac","frac_issue","avg_kernel_ms","R_eff","timed_where")}, r["composite_fwd"]["avg_kernel_ms"], r["composite_fwd"]["frac"])
print(d["cpu_baseline"]["value"], d["iters_per_sec_1k"], d["fps_reference_method"]["fps"])
PY
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_scene_io_gpu.py -x -q -m gpu -s -k "left_and_reentered or commit_gate or fused_synced or scene_trains" > gpurun_out/r4/tests_v2.log 2>&1; echo "tests rc $?"; grep -E "PSNR|passed|failed" gpurun_out/r4/tests_v2.log | tail
