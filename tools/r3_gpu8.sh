#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash tools/trace_seq.sh | grep -E "fwd|bwd" | grep -v "true>"
bash tools/prof.sh r03_c4_1M_1080p python tools/c4_probe.py > /dev/null 2>&1; head -4 gpurun_out/r03_c4_1M_1080p_kernel_stats.csv | cut -c1-60,200-330
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -2
