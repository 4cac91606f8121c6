#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
for d in 0 3; do timeout 300 python bench.py --sh-degree $d --cpu-iters 0 --no-long-run 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sh $d', {k:round(v['iters_per_sec']) for k,v in d['loops'].items()})"; done
bash tools/prof.sh r4_sh3 python bench.py --sh-degree 3 --steps 20 --warmup 5 --cpu-iters 0 --no-long-run > /dev/null
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/r4_sh3_kernel_stats.csv")):
    if int(r["Calls"])>200: print("  %6s avg %8.2f us  %s" % (r["Calls"], float(r["AverageNs"])/1e3, r["Name"].replace("(anonymous namespace)::","").replace("void ","")[:90]))
PY
