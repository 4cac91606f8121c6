#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -3
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), {k:round(v['iters_per_sec']) for k,v in d['loops'].items()})"; done
bash tools/prof.sh r4_lk2 python tools/loop_kernels.py run_ahead 400 > /dev/null
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/r4_lk2_kernel_stats.csv")):
    c=int(r["Calls"])
    if c>=40: print("%5.2f/iter avg %7.2f us  %s" % (c/400.0, float(r["AverageNs"])/1e3, r["Name"].replace("(anonymous namespace)::","")[:80]))
PY
