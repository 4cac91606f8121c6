#!/bin/bash
# usage: tools_pmc.sh <name> <counter> <cmd...>  -> gpurun_out/<name>_<counter>.csv (per-kernel mean of the counter)
name=$1; ctr=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_${name}_${ctr}
mkdir -p $out
timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o pmc -- "$@" > $out.log 2>&1 < /dev/null
f=$(find $out -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then
python3 - "$f" "$ctr" > gpurun_out/pmc_${name}_${ctr}.csv <<'PY'
import csv, sys, collections
f, ctr = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
with open(f) as fh:
    for r in csv.DictReader(fh):
        if r.get("Counter_Name") != ctr: continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
print("kernel,dispatches,mean_%s" % ctr)
for k, (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%s,%d,%.3f" % (k.replace(",", ";"), n, s / n))
PY
head -12 gpurun_out/pmc_${name}_${ctr}.csv
else
tail -5 $out.log
fi
rm -rf $out
