#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
bash tools/prof.sh r4_lk python tools/loop_kernels.py dropin 400 > /dev/null
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/r4_lk_kernel_stats.csv")):
    c=int(r["Calls"])
    if c>=150: print("%5.2f/iter avg %7.2f us  %s" % (c/400.0, float(r["AverageNs"])/1e3, r["Name"].replace("(anonymous namespace)::","")[:230]))
PY
