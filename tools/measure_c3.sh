# the C3 part of tools/measure_round.sh (bench line, kernel stats, three counter passes)      usage: bash tools/measure_c3.sh r03
R=${1:-rXX}
cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py > gpurun_out/${R}_bench_default_run.json 2> gpurun_out/bench_default.err; tail -c 300 gpurun_out/bench_default.err
bash tools/prof.sh ${R}_bench_c3 python bench.py --steps 200 --warmup 20 --cpu-iters 0 --no-long-run > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_ACTIVE_INST_VALU,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAVES; do
  bash tools/pmc.sh c3 $c python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run > /dev/null 2>&1
done
cp gpurun_out/pmc_c3_FETCH_SIZE.csv gpurun_out/${R}_pmc_c3_FETCH_SIZE.csv; cp gpurun_out/pmc_c3_WRITE_SIZE.csv gpurun_out/${R}_pmc_c3_WRITE_SIZE.csv
cp gpurun_out/pmc_c3_SQ_INSTS_VALU_SQ_INSTS_SALU_SQ_INSTS_LDS_SQ_ACTIVE_INST_VALU_SQ_WAVE_CYCLES_SQ_BUSY_CYCLES_SQ_WAVES.csv gpurun_out/${R}_pmc_c3_SQ_counters.csv
timeout 300 python bench.py --gpus 2 --steps 100 --cpu-iters 0 2> gpurun_out/bench_gpus2.err | grep "^{" > gpurun_out/${R}_bench_gpus2_shared_gpu.json; tail -c 200 gpurun_out/bench_gpus2.err
head -12 gpurun_out/${R}_bench_c3_kernel_stats.csv | cut -c1-90
