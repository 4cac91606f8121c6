cd "$GRAFT_REPO_ROOT"
timeout 300 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 300 gpurun_out/bench_default.err
tools/prof.sh v10 python bench.py --steps 200 --warmup 20 --cpu-iters 0 > /dev/null 2>&1
tools/pmc.sh c3 FETCH_SIZE python bench.py --steps 20 --warmup 5 --cpu-iters 0 > /dev/null 2>&1
tools/pmc.sh c3 WRITE_SIZE python bench.py --steps 20 --warmup 5 --cpu-iters 0 > /dev/null 2>&1
tools/pmc.sh c3 SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_ACTIVE_INST_VALU,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAVES python bench.py --steps 20 --warmup 5 --cpu-iters 0 > /dev/null 2>&1
timeout 300 python tools/c4_probe.py > gpurun_out/c4_probe.txt 2>&1; tail -12 gpurun_out/c4_probe.txt
ls -la gpurun_out | tail -12
