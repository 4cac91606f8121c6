# Round measurements on the GPU box (gpurun): default bench line, kernel stats, three separate PMC passes (counters never share a
# run with other trace domains), the C4 probe and the full GPU test tier LAST.  Summaries land in gpurun_out/; the ones to be
# judged are copied into profiles/ by hand (named per round).
cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 300 gpurun_out/bench_default.err
bash tools/prof.sh rXX python bench.py --steps 200 --warmup 20 --cpu-iters 0 --no-long-run > /dev/null 2>&1
bash tools/pmc.sh c3 FETCH_SIZE python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run > /dev/null 2>&1
bash tools/pmc.sh c3 WRITE_SIZE python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run > /dev/null 2>&1
bash tools/pmc.sh c3 SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_ACTIVE_INST_VALU,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAVES python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run > /dev/null 2>&1
timeout 300 python tools/c4_probe.py > gpurun_out/c4_probe.txt 2>&1; tail -3 gpurun_out/c4_probe.txt
python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/gputest_last.txt; cat gpurun_out/gputest_last.txt
ls gpurun_out | tail -12
