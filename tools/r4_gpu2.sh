#!/bin/bash
# round 4, GPU call 2: the new scene-from-directory GPU test, then the whole GPU tier
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_scene_io_gpu.py -x -q -m gpu -s > gpurun_out/r4/scene_io_gpu.log 2>&1; echo "scene_io rc $?"
grep -E "PSNR|passed|failed|Error|assert" gpurun_out/r4/scene_io_gpu.log | head -20
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4/gpu_tier.log 2>&1; echo "gpu tier rc $?"
tail -5 gpurun_out/r4/gpu_tier.log
