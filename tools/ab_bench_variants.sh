#!/bin/bash
# GPU box: tools/ab_bench_variants.sh <variant>...   alternates the library builds under instantsplat_amd/lib/variants/ (tools/build_variant.sh)
# and prints, per run, the bench's own numbers: composite kernel times (HIP events, untimed pass) and the loops' rates
cd "$GRAFT_REPO_ROOT"
L=instantsplat_amd/lib
cp $L/libmi355gs.so /tmp/keep.so
for rep in 1 2 3; do for v in "$@"; do
  cp $L/variants/$v.so $L/libmi355gs.so
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('%-10s bwd %.2f us  fwd %.2f us  | dropin %.0f  synced %.0f  run-ahead %.0f it/s' % ('$v', r['avg_kernel_ms']*1e3, r['composite_fwd']['avg_kernel_ms']*1e3, d['value'], d['iters_per_sec_one_call_synced'], d['iters_per_sec_run_ahead']))"
done; done
last="${@: -1}"
cp $L/variants/$last.so $L/libmi355gs.so
timeout 900 python -m pytest ${TESTS:-tests/test_raster_gpu.py tests/test_edge_gpu.py tests/test_properties_gpu.py tests/test_baseline_sizes_gpu.py tests/test_zz_reference_functions_gpu.py} -x -q 2>&1 | tail -3
cp /tmp/keep.so $L/libmi355gs.so
