#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4f
for loop in one_call dropin; do
  rm -rf gpurun_out/gap_$loop; mkdir -p gpurun_out/gap_$loop
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap_$loop -o t -- python tools/loop_kernels.py $loop 400 > gpurun_out/gap_$loop.log 2>&1 < /dev/null
  f=$(find gpurun_out/gap_$loop -name "*kernel_trace.csv" | head -1)
  echo "== $loop"; python tools/gap_analysis.py "$f" 300 | tee gpurun_out/r4f/gaps_$loop.txt | head -12
  rm -rf gpurun_out/gap_$loop
done
