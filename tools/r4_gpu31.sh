#!/bin/bash
# round 4, final tree: the PMC passes again (separate --pmc runs, rocprofv3 --kernel-trace only) for C3 (the bench command) and C4
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4f
BENCH="python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run"
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES; do
  bash tools/pmc.sh c3 $c $BENCH | head -4
  bash tools/pmc.sh c4 $c python tools/c4_probe.py | head -4
done
ls gpurun_out/pmc_c3_*.csv gpurun_out/pmc_c4_*.csv
