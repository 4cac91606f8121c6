cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/c2_probe2.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c2_probe2.txt
timeout 600 python tools/onek_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_onek_probe.txt
