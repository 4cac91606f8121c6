#!/bin/bash
cd "$GRAFT_REPO_ROOT"
GS_CALIBRATE=1 timeout 300 python -m pytest tests/test_ops_gpu.py -q -k "compiled_binding" 2>&1 | grep "CAL compiled" | sort -k5 -g | tail -8
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | tail -6
MI355GS_BINDING=compiled timeout 300 python tools/host_timeline.py 600 2>&1 | grep -E "drop-in|render\(\)|optimizer.step"
