cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for s in 31 32; do timeout 900 python tools/fuzz_raster_emu.py $s 160 gpu 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/r06_fuzz_gpu_seed$s.txt; tail -3 gpurun_out/r06_fuzz_gpu_seed$s.txt; done
timeout 900 python tools/fuzz_ops.py 41 150 gpu 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r06_fuzz_ops_gpu_seed41.txt; cat gpurun_out/r06_fuzz_ops_gpu_seed41.txt
timeout 900 python tools/soak.py 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/r06_soak_3100_iterations.txt; cut -c1-600 gpurun_out/r06_soak_3100_iterations.txt
