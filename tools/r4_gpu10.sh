#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for l in dropin one_call; do
  bash tools/prof.sh r4_loop_$l python tools/loop_kernels.py $l 400 > /dev/null
  python - gpurun_out/r4_loop_${l}_kernel_stats.csv $l <<'PY'
import csv, sys
print("====", sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    c = int(r["Calls"])
    if c >= 100: print("  %6.2f / iter  avg %8.2f us  %s" % (c / 400.0, float(r["AverageNs"]) / 1e3, r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:110]))
PY
done
