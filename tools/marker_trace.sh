#!/bin/bash
# usage: tools/marker_trace.sh <name> <cmd...> -> gpurun_out/<name>_marker_ranges.txt: the library's roctx ranges (MI355GS_ROCTX=1,
# include/mi355gs.h mi355gs_profile_ranges) of <cmd> under `rocprofv3 --marker-trace --kernel-trace` (no counters in this pass),
# summarised per range name: count, mean host microseconds.
name=$1; shift
cd /tmp && export TMPDIR=/tmp
export MI355GS_BENCH_CHILD=1 MI355GS_ROCTX=1
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/marker_$name
mkdir -p $out
timeout 600 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $out -o $name -- "$@" > $out.log 2>&1 < /dev/null
f=$(find $out -name "*marker_api_trace.csv" | head -1)
if [ -n "$f" ]; then
python3 - "$f" > gpurun_out/${name}_marker_ranges.txt <<'PY'
import csv, sys, collections
acc = collections.OrderedDict()
with open(sys.argv[1]) as fh:
    rows = list(csv.DictReader(fh))
cols = rows[0].keys() if rows else []
for r in rows:
    n = r.get("Function") or r.get("Name") or "?"
    a = acc.setdefault(n, [0, 0.0])
    a[0] += 1; a[1] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
print("# rocprofv3 --marker-trace --kernel-trace, MI355GS_ROCTX=1; columns of the trace: " + ", ".join(cols))
print("%-44s %8s %12s" % ("range", "count", "mean host us"))
for n, (c, t) in acc.items():
    print("%-44s %8d %12.2f" % (n, c, t / max(c, 1)))
PY
cat gpurun_out/${name}_marker_ranges.txt
else
tail -5 $out.log; find $out -type f | head
fi
rm -rf $out
