#!/bin/bash
# round 4, GPU call 8: binning after the wide-queue change (C4 + C3 kernel stats), binning tests, bench (forward issue model fixed)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
bash tools/prof.sh r4_c4 python tools/c4_probe.py > /dev/null
bash tools/prof.sh r4_c3 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run > /dev/null
timeout 900 python -m pytest tests/test_edge_gpu.py tests/test_raster_gpu.py tests/test_baseline_sizes_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run > gpurun_out/r4/bench_v4.json 2> gpurun_out/r4/bench_v4.err; echo "bench rc $?"
