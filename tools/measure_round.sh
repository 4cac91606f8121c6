# One GPU call's worth of round records (gpurun): the GPU test tier, the default bench line, rocprofv3 kernel stats and four separate
# PMC passes of the bench command (traffic, SQ issue counters, SQ wait counters — counters never share a run with other trace
# domains), the render-only A/B with its traffic passes, the deterministic mode's cost and kernels, the other BASELINE configs.
# Summaries land in gpurun_out/ under the names they are copied to profiles/ with.      usage: bash tools/measure_round.sh r05
R=${1:-rXX}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/${R}_bench_default_run_v3.json 2> gpurun_out/bench_default.err; tail -c 400 gpurun_out/bench_default.err; cp gpurun_out/bench_full_n1.json gpurun_out/${R}_bench_default_full_record.json
bash tools/prof.sh ${R}_bench_c3 python bench.py --steps 200 --warmup 20 --cpu-iters 0 --no-long-run --no-other-configs > /dev/null 2>&1
A=SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES
B=SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU,SQ_BUSY_CYCLES,SQ_WAVES
for c in FETCH_SIZE WRITE_SIZE $A $B; do
  bash tools/pmc.sh c3 $c python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run --no-other-configs > /dev/null 2>&1
done
cp gpurun_out/pmc_c3_FETCH_SIZE.csv gpurun_out/${R}_pmc_c3_FETCH_SIZE.csv; cp gpurun_out/pmc_c3_WRITE_SIZE.csv gpurun_out/${R}_pmc_c3_WRITE_SIZE.csv
cp gpurun_out/pmc_c3_${A//,/_}.csv gpurun_out/${R}_pmc_c3_SQ_counters.csv; cp gpurun_out/pmc_c3_${B//,/_}.csv gpurun_out/${R}_pmc_c3_SQ_wait_counters.csv
head -14 gpurun_out/${R}_pmc_c3_SQ_wait_counters.csv | cut -c1-200
timeout 900 python tools/configs.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_baseline_configs_c1_c2_c4.txt; cat gpurun_out/${R}_baseline_configs_c1_c2_c4.txt
timeout 300 python tools/render_only_loop.py 2>/dev/null | grep "^{" > gpurun_out/${R}_ab_render_only_forward.txt; cat gpurun_out/${R}_ab_render_only_forward.txt
for c in FETCH_SIZE WRITE_SIZE; do
  bash tools/pmc.sh render_only $c python tools/render_only_loop.py 10 > /dev/null 2>&1
  cp gpurun_out/pmc_render_only_$c.csv gpurun_out/${R}_pmc_render_only_$c.csv; grep -E "kernel,|composite_fwd" gpurun_out/${R}_pmc_render_only_$c.csv
done
bash tools/prof.sh ${R}_c4_1M_1080p python tools/c4_probe.py > /dev/null 2>&1
timeout 300 python tools/c4_probe.py 2>&1 | grep "^C4" > gpurun_out/${R}_c4_probe.txt; cat gpurun_out/${R}_c4_probe.txt
for c in FETCH_SIZE WRITE_SIZE; do
  GS_C4_DET=0 bash tools/pmc.sh c4 $c python tools/c4_probe.py > /dev/null 2>&1
  cp gpurun_out/pmc_c4_$c.csv gpurun_out/${R}_pmc_c4_$c.csv
done
python tools/kernel_roofline_table.py gpurun_out/${R}_bench_c3_kernel_stats.csv gpurun_out/${R}_pmc_c3 > gpurun_out/${R}_kernel_traffic_table.txt 2>&1; head -20 gpurun_out/${R}_kernel_traffic_table.txt
python tools/kernel_roofline_table.py gpurun_out/${R}_c4_1M_1080p_kernel_stats.csv gpurun_out/${R}_pmc_c4 10 > gpurun_out/${R}_kernel_traffic_table_c4.txt 2>&1
timeout 300 python tools/det_cost.py 2>/dev/null | grep "^{" > gpurun_out/${R}_deterministic_mode_cost.txt; cat gpurun_out/${R}_deterministic_mode_cost.txt
bash tools/prof.sh ${R}_det_c3 python tools/det_cost.py 100 > /dev/null 2>&1
R=$R python - <<'PY'
import json, os
d = json.load(open("gpurun_out/%s_bench_default_run_v3.json" % os.environ["R"]))
r = d["roofline"]
print("value %.0f | loops %s" % (d["value"], {k: round(v) for k, v in d["loops"].items()}))
print({k: (round(v["avg_kernel_ms"] * 1e3, 1), round(v["frac"], 3)) for k, v in r["small_kernels"].items()})
PY
ls gpurun_out | grep ${R}_
