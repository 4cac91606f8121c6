#!/bin/bash
# round 4, GPU call: records for profiles/ — C4 PMC passes, SH degree 3 line, two ranks on the one GPU, host timeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
bash tools/pmc.sh c4 FETCH_SIZE python tools/c4_probe.py | head -6
bash tools/pmc.sh c4 WRITE_SIZE python tools/c4_probe.py | head -6
bash tools/pmc.sh c4 SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES python tools/c4_probe.py | head -6
timeout 600 python bench.py --sh-degree 3 --cpu-iters 0 --no-long-run > gpurun_out/r4/bench_sh3.json 2> gpurun_out/r4/bench_sh3.err; echo "sh3 rc $?"
timeout 900 python bench.py --gpus 2 --steps 100 --cpu-iters 0 > gpurun_out/r4/bench_gpus2.json 2> gpurun_out/r4/bench_gpus2.err; echo "gpus2 rc $?"; tail -3 gpurun_out/r4/bench_gpus2.err
MI355GS_BINDING=compiled timeout 300 python tools/host_timeline.py 600 > gpurun_out/r4/host_timeline_compiled.txt 2>&1; tail -20 gpurun_out/r4/host_timeline_compiled.txt
python - <<'PY'
import json
for f in ("gpurun_out/r4/bench_sh3.json","gpurun_out/r4/bench_gpus2.json"):
    try:
        d=json.load(open(f)); print(f, d["n_gpus"], round(d["value"]), {k:round(v["iters_per_sec"]) for k,v in d["loops"].items()}, d.get("legs_skipped"))
        if d.get("multi_gpu"): print("   ", d["multi_gpu"]["backend"], d["multi_gpu"]["solo_rank0_iters_per_sec"], d["multi_gpu"]["scaling_efficiency_vs_solo_rank0"], [round(r["iters_per_sec_median_block_own_clock"]) for r in d["multi_gpu"]["per_rank"]])
    except Exception as e: print(f, "ERR", e)
PY
