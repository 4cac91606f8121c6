#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
loop=${1:-run_ahead}
rm -rf gpurun_out/gap_$loop; mkdir -p gpurun_out/gap_$loop
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/gap_$loop -o t -- python tools/loop_kernels.py $loop 400 > gpurun_out/gap_$loop.log 2>&1 < /dev/null
ls gpurun_out/gap_$loop
f=$(find gpurun_out/gap_$loop -name "*kernel_trace.csv" | head -1)
head -2 "$f" | cut -c1-600
python - "$f" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
bwd=[i for i,r in enumerate(rows) if "k_composite_bwd" in r["Kernel_Name"]]
a=bwd[-100]; t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:a+32]:
    print("%9.2f %9.2f  q%s  %s" % ((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3, r.get("Queue_Id","?"), r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:50]))
PY
m=$(find gpurun_out/gap_$loop -name "*memory_copy_trace.csv" | head -1); [ -n "$m" ] && { head -3 "$m" | cut -c1-300; wc -l "$m"; }
rm -rf gpurun_out/gap_$loop
