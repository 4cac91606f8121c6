#!/bin/bash
# usage: tools_prof.sh <name> <cmd...>   -> gpurun_out/<name>_kernel_stats.csv (+ trace)
name=$1; shift
cd /tmp && export TMPDIR=/tmp
export MI355GS_BENCH_CHILD=1   # bench.py measures in this process (no supervising parent): rocprofv3 sees the process that launches the kernels
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$name
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$name -o $name -- "$@" > gpurun_out/$name.log 2>&1 < /dev/null
find gpurun_out/$name -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} gpurun_out/${name}_kernel_stats.csv
tail -4 gpurun_out/$name.log
[ -f gpurun_out/${name}_kernel_stats.csv ] && head -16 gpurun_out/${name}_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/$name   # traces are big; keep the summary
