#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4f
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_zz_reference_functions_gpu.py -x -q 2>&1 | tail -2
for b in compiled ctypes; do MI355GS_BINDING=$b timeout 600 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$b', round(d['value']), {k:round(v['iters_per_sec']) for k,v in d['loops'].items()})"; done
