cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for i in 1 2; do timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu | tail -2; done
timeout 600 python -m pytest tests/test_baseline_sizes_gpu.py -q -s -k "deterministic_mode_two_runs" 2>&1 | grep -v amdgpu | grep "deterministic mode"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu
