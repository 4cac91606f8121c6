#!/bin/bash
# same-box A/B of library variants on the C3 bench under rocprofv3: kernels matching $GREP
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
L=instantsplat_amd/lib; cp $L/libmi355gs.so /tmp/keep.so
for rep in 1 2; do for v in "$@"; do
  cp $L/variants/$v.so $L/libmi355gs.so
  bash tools/prof.sh r4_ab_${v} python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run > /dev/null
  python - "$v" "${GREP:-adam}" <<'PY'
import csv,sys
for r in csv.DictReader(open("gpurun_out/r4_ab_%s_kernel_stats.csv" % sys.argv[1])):
    if sys.argv[2] in r["Name"] and int(r["Calls"])>100: print("%-12s %6s avg %8.2f us  %s" % (sys.argv[1], r["Calls"], float(r["AverageNs"])/1e3, r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:40]))
PY
done; done
cp /tmp/keep.so $L/libmi355gs.so
