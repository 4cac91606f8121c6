#!/bin/bash
# round 4, GPU call: dense depth array for the scatter — kernel stats at C3 (bench) and C4 (probe)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
show() { python - "$1" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if any(k in n for k in ("scatter","scan","count","sort","preprocess","composite")): print("  %6s avg %8.2f us  %s" % (r["Calls"], float(r["AverageNs"])/1e3, n.replace("(anonymous namespace)::","").replace("void ","")[:80]))
PY
}
bash tools/prof.sh r4_depth_c4 python tools/c4_probe.py > /dev/null; show gpurun_out/r4_depth_c4_kernel_stats.csv
bash tools/prof.sh r4_depth_c3 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run > /dev/null; show gpurun_out/r4_depth_c3_kernel_stats.csv
timeout 600 python bench.py --cpu-iters 0 --no-long-run 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), {k:round(v['iters_per_sec']) for k,v in d['loops'].items()})"
timeout 900 python -m pytest tests -m gpu -x -q -k "raster or edge or baseline" 2>&1 | tail -3
