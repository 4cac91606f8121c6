#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
for v in base half; do echo "== probe build: $v (GS_BW_HALF=$([ $v = half ] && echo 1 || echo 0))"; GS_PROBE_LIB=instantsplat_amd/lib/variants/probe_$v.so timeout 300 python tools/probe_bwd.py 200 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r4/bwd_probe_base_vs_half.txt
cat gpurun_out/r4/bwd_probe_base_vs_half.txt
