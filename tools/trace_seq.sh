#!/bin/bash
# usage (GPU box): tools/trace_seq.sh  -> gpurun_out/tr_avg.txt: per-kernel average duration over a short bench run
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/tr
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -o tr -- python bench.py --cpu-iters 0 --steps 60 --warmup 10 > gpurun_out/tr.log 2>&1 < /dev/null
f=$(find gpurun_out/tr -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:28]
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = []
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    out.append(f"{n:28s} n={len(v):4d} avg={sum(v)/len(v):7.1f} first={v[0]:6.1f} last={v[-1]:6.1f}")
open("gpurun_out/tr_avg.txt", "w").write("\n".join(out[:24]))
print("\n".join(out[:22]))
PY
rm -rf gpurun_out/tr
