cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python tools/diag_train_case.py gpu 2>&1 | grep "^case" | tee gpurun_out/r06_diag_train_case_seed41.txt | cut -c1-420
timeout 900 python tools/fuzz_ops.py 41 150 gpu 2>&1 | grep -E "^FAIL|^seed" | tee gpurun_out/r06_fuzz_ops_gpu_seed41.txt
timeout 900 python tools/fuzz_ops.py 42 150 gpu 2>&1 | grep -E "^FAIL|^seed" | tee gpurun_out/r06_fuzz_ops_gpu_seed42.txt
