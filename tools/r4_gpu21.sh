#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
timeout 600 python tools/ab_dropin_housekeeping.py 1500 2>/dev/null | tee gpurun_out/r4/ab_dropin_housekeeping.txt
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_scene_io_gpu.py -x -q 2>&1 | tail -3
