#!/bin/bash
# timing-only experiments on the C4 probe: library variants under rocprofv3
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
L=instantsplat_amd/lib; cp $L/libmi355gs.so /tmp/keep.so
show() { python - "$1" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if any(k in n for k in ("scatter_lds","scan_tiles","count_tiles","sort_tiles")) and int(r["Calls"])>20: print("  %6s avg %8.2f us  %s" % (r["Calls"], float(r["AverageNs"])/1e3, n.replace("(anonymous namespace)::","").replace("void ","")[:40]))
PY
}
for v in "$@"; do
  cp $L/variants/$v.so $L/libmi355gs.so
  echo "== $v C4"; bash tools/prof.sh r4_ab_${v}_c4 python tools/c4_probe.py > /dev/null; show gpurun_out/r4_ab_${v}_c4_kernel_stats.csv
done
cp /tmp/keep.so $L/libmi355gs.so
