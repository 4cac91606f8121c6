#!/bin/bash
# which kind of box is this, and what do the step's kernels take on it
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4h
python tools/box_probe.py 2>&1 | tee gpurun_out/r4h/box_probe.txt
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('== bench: bwd %.2f us  fwd %.2f us  | dropin %.0f  synced %.0f  run-ahead %.0f it/s' % (r['avg_kernel_ms']*1e3, r['composite_fwd']['avg_kernel_ms']*1e3, d['value'], d['iters_per_sec_one_call_synced'], d['iters_per_sec_run_ahead']))" | tee -a gpurun_out/r4h/box_probe.txt
