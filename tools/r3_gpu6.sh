#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_zz_reference_functions_gpu.py tests/test_raster_gpu.py -q --tb=short 2>&1 | grep -v "^$" | tail -40
MI355GS_BINDING=ctypes timeout 600 python -m pytest tests/test_zz_reference_functions_gpu.py -q --tb=line 2>&1 | tail -5
