#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4f
timeout 900 python bench.py --gpus 2 --steps 100 --cpu-iters 0 > gpurun_out/r4f/bench_gpus2.json 2> gpurun_out/r4f/bench_gpus2.err; echo "gpus2 rc $?"; tail -3 gpurun_out/r4f/bench_gpus2.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4f/bench_gpus2.json")); print(d["n_gpus"], round(d["value"]), {k:round(v["iters_per_sec"]) for k,v in d["loops"].items()}, d.get("legs_skipped"))
m=d["multi_gpu"]; print(m["backend"], m["solo_rank0_iters_per_sec"], m["scaling_efficiency_vs_solo_rank0"], [round(r["iters_per_sec_median_block_own_clock"]) for r in m["per_rank"]])
PY
timeout 600 python -m pytest tests/test_rccl_gpu.py -m gpu -q 2>&1 | tail -2
