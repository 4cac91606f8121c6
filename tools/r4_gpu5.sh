#!/bin/bash
# round 4, GPU call 5: binning at C4 (staged tile scan, tile-box passes) and at C3 — kernel stats under rocprofv3
cd "$GRAFT_REPO_ROOT"
bash tools/prof.sh r4_c4 python tools/c4_probe.py
bash tools/prof.sh r4_c3 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run
timeout 900 python -m pytest tests/test_baseline_sizes_gpu.py tests/test_raster_gpu.py tests/test_edge_gpu.py -x -q -m gpu 2>&1 | tail -3
