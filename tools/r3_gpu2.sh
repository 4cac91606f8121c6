#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "compiled" 2>&1 | grep -v "^$" | tail -30
MI355GS_BINDING=compiled timeout 300 python tools/host_timeline.py 600 > gpurun_out/r03_dropin_host_timeline_compiled.txt 2>&1; cat gpurun_out/r03_dropin_host_timeline_compiled.txt
GS_SINGLE_THREAD_AUTOGRAD=1 MI355GS_BINDING=compiled timeout 300 python tools/host_timeline.py 600 > gpurun_out/r03_dropin_host_timeline_compiled_st.txt 2>&1; cat gpurun_out/r03_dropin_host_timeline_compiled_st.txt
timeout 600 python tools/configs.py > gpurun_out/r03_baseline_configs.txt 2>&1; cat gpurun_out/r03_baseline_configs.txt | grep -v amdgpu.ids
