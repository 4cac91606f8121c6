cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 python tools/c2_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c2_probe.txt
GS_AB_TESTS=1 bash tools/ab_quick.sh 3 base addtid pair pairaddtid 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_ab_bwd_packed_pairs.txt
