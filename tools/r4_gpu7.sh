#!/bin/bash
# round 4, GPU call 7: full GPU tier, trained-state ratio distribution, the other BASELINE configs, the bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tier_v2.log 2>&1; echo "gpu tier rc $?"; tail -3 gpurun_out/r4/gpu_tier_v2.log
timeout 1200 python tools/trained_state_ratios.py 30 6 > gpurun_out/r4/trained_state_ratios.txt 2>&1; echo "ratios rc $?"; grep -E "^C3|^C4" gpurun_out/r4/trained_state_ratios.txt | cut -c1-160
timeout 900 python tools/configs.py > gpurun_out/r4/baseline_configs.txt 2>&1; echo "configs rc $?"; tail -12 gpurun_out/r4/baseline_configs.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_v3_driver.json 2> gpurun_out/r4/bench_v3_driver.err; echo "bench rc $?"
timeout 600 python bench.py > gpurun_out/r4/bench_v3_default.json 2> gpurun_out/r4/bench_v3_default.err; echo "bench default rc $?"
