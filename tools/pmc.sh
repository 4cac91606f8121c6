#!/bin/bash
# usage: tools_pmc.sh <name> <counter> <cmd...>  -> gpurun_out/<name>_<counter>.csv (per-kernel mean of the counter)
name=$1; ctr=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
export MI355GS_BENCH_CHILD=1   # bench.py measures in this process (no supervising parent): rocprofv3 sees the process that launches the kernels
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_${name}_${ctr//,/_}
mkdir -p $out
timeout 600 rocprofv3 --pmc ${ctr//,/ } --kernel-trace --output-format csv -d $out -o pmc -- "$@" > $out.log 2>&1 < /dev/null
f=$(find $out -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then
python3 - "$f" "$ctr" > gpurun_out/pmc_${name}_${ctr//,/_}.csv <<'PY'
import csv, sys, collections
f, ctr = sys.argv[1], sys.argv[2]
ctrs = ctr.split(",")
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
with open(f) as fh:
    for r in csv.DictReader(fh):
        c = r.get("Counter_Name")
        if c not in ctrs: continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][c][0] += 1; acc[k][c][1] += float(r["Counter_Value"])
print("kernel,dispatches," + ",".join("mean_%s" % c for c in ctrs))
for k, d in sorted(acc.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
    n = max(v[0] for v in d.values())
    print("%s,%d,%s" % (k.replace(",", ";"), n, ",".join("%.3f" % (d[c][1] / max(d[c][0], 1)) for c in ctrs)))
PY
head -12 gpurun_out/pmc_${name}_${ctr//,/_}.csv
else
tail -5 $out.log
fi
rm -rf $out
