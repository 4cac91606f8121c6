#!/bin/bash
# round 4, GPU call 6: issue-model validation (fwd + bwd) against SQ_INSTS_VALU, PMC passes of the bench command, kernel stats
cd "$GRAFT_REPO_ROOT"
BENCH="python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run"
bash tools/pmc.sh issue_model SQ_INSTS_VALU,SQ_INSTS_LDS python tools/validate_issue_model.py > gpurun_out/r4_issue_model_pmc.txt 2>&1
grep -h "^{" gpurun_out/pmc_issue_model_SQ_INSTS_VALU_SQ_INSTS_LDS.log >> gpurun_out/r4_issue_model_pmc.txt
cat gpurun_out/r4_issue_model_pmc.txt | cut -c1-400
bash tools/pmc.sh c3 FETCH_SIZE $BENCH | head -8
bash tools/pmc.sh c3 WRITE_SIZE $BENCH | head -8
bash tools/pmc.sh c3 SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES $BENCH | head -8
bash tools/prof.sh r4_c3 $BENCH > /dev/null
bash tools/prof.sh r4_c4 python tools/c4_probe.py > /dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench_v3_driver.json 2> gpurun_out/r4/bench_v3_driver.err; echo "bench rc $?"
