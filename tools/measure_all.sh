# Round measurements on the GPU box (gpurun): default bench line, kernel stats, three separate PMC passes at C3 and at C4 (counters
# never share a run with other trace domains), the VALU-issue model checked against SQ_INSTS_VALU on fixed frames, the other
# BASELINE configs, the 2-rank path on the shared GPU, and the full GPU test tier LAST.  Summaries land in gpurun_out/ under
# the names they are copied to profiles/ with.      usage: bash tools/measure_all.sh r03
R=${1:-rXX}
cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py > gpurun_out/${R}_bench_default_run.json 2> gpurun_out/bench_default.err; tail -c 300 gpurun_out/bench_default.err
bash tools/prof.sh ${R}_bench_c3 python bench.py --steps 200 --warmup 20 --cpu-iters 0 --no-long-run > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_ACTIVE_INST_VALU,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAVES; do
  bash tools/pmc.sh c3 $c python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run > /dev/null 2>&1
done
cp gpurun_out/pmc_c3_FETCH_SIZE.csv gpurun_out/${R}_pmc_c3_FETCH_SIZE.csv; cp gpurun_out/pmc_c3_WRITE_SIZE.csv gpurun_out/${R}_pmc_c3_WRITE_SIZE.csv
cp gpurun_out/pmc_c3_SQ_INSTS_VALU_SQ_INSTS_SALU_SQ_INSTS_LDS_SQ_ACTIVE_INST_VALU_SQ_WAVE_CYCLES_SQ_BUSY_CYCLES_SQ_WAVES.csv gpurun_out/${R}_pmc_c3_SQ_counters.csv
# the issue model against the counter, same frames
bash tools/pmc.sh issue_model SQ_INSTS_VALU,SQ_INSTS_LDS python tools/validate_issue_model.py > /dev/null 2>&1
{ grep -h "^{" gpurun_out/pmc_issue_model_SQ_INSTS_VALU_SQ_INSTS_LDS.log; grep -E "kernel,|composite_bwd" gpurun_out/pmc_issue_model_SQ_INSTS_VALU_SQ_INSTS_LDS.csv; } > gpurun_out/${R}_issue_model_vs_SQ_INSTS_VALU.txt
cat gpurun_out/${R}_issue_model_vs_SQ_INSTS_VALU.txt
# C4
bash tools/prof.sh ${R}_c4_1M_1080p python tools/c4_probe.py > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_ACTIVE_INST_VALU,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAVES; do
  bash tools/pmc.sh c4 $c python tools/c4_probe.py > /dev/null 2>&1
done
cp gpurun_out/pmc_c4_FETCH_SIZE.csv gpurun_out/${R}_pmc_c4_FETCH_SIZE.csv; cp gpurun_out/pmc_c4_WRITE_SIZE.csv gpurun_out/${R}_pmc_c4_WRITE_SIZE.csv
cp gpurun_out/pmc_c4_SQ_INSTS_VALU_SQ_INSTS_SALU_SQ_INSTS_LDS_SQ_ACTIVE_INST_VALU_SQ_WAVE_CYCLES_SQ_BUSY_CYCLES_SQ_WAVES.csv gpurun_out/${R}_pmc_c4_SQ_counters.csv
timeout 600 python tools/configs.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_baseline_configs_c1_c2_c4.txt; cat gpurun_out/${R}_baseline_configs_c1_c2_c4.txt
timeout 300 python bench.py --gpus 2 --steps 100 --cpu-iters 0 2> gpurun_out/bench_gpus2.err | grep "^{" > gpurun_out/${R}_bench_gpus2_shared_gpu.json; tail -c 200 gpurun_out/bench_gpus2.err   # (gloo announces its connections on stdout: keep the JSON line only)
MI355GS_BINDING=compiled timeout 300 python tools/host_timeline.py 600 2>&1 | grep -v amdgpu > gpurun_out/${R}_dropin_host_timeline_compiled.txt
MI355GS_BINDING=ctypes timeout 300 python tools/host_timeline.py 600 2>&1 | grep -v amdgpu > gpurun_out/${R}_dropin_host_timeline_ctypes.txt
head -3 gpurun_out/${R}_dropin_host_timeline_compiled.txt gpurun_out/${R}_dropin_host_timeline_ctypes.txt
python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/${R}_gputest_tail.txt; cat gpurun_out/${R}_gputest_tail.txt
ls gpurun_out | grep ${R}_
