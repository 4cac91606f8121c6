cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for s in 41 42 43; do timeout 900 python tools/fuzz_ops.py $s 150 gpu 2>&1 | grep -E "^FAIL|^seed" | tee gpurun_out/r06_fuzz_ops_gpu_seed$s.txt; done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r06_gputest_log_v3.txt; cat gpurun_out/r06_gputest_log_v3.txt
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_args_run.json 2> gpurun_out/bench.err ) 2>&1 | tail -3; wc -c gpurun_out/r06_bench_driver_args_run.json; tail -c 300 gpurun_out/bench.err
cp gpurun_out/bench_full_n1.json gpurun_out/r06_bench_driver_args_full.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
