#!/bin/bash
# GPU box: tools/ab_quick.sh <reps> <variant>...  alternates library builds (tools/build_variant.sh), bench's own kernel times and loop rates
cd "$GRAFT_REPO_ROOT"
# GS_AB_TESTS=1: every variant first runs the rasterizer's GPU parity tests (an A/B of a build that is wrong is not a result)
L=instantsplat_amd/lib
python tools/box_probe.py 2>&1 | grep -E "device copy|v_fma_f32 +waves/SIMD 4"
cp $L/libmi355gs.so /tmp/keep.so
reps=$1; shift
if [ "$GS_AB_TESTS" = 1 ]; then for v in "$@"; do
  cp $L/variants/$v.so $L/libmi355gs.so
  echo "$v: $(timeout 900 python -m pytest tests/test_raster_gpu.py tests/test_baseline_sizes_gpu.py -q -x 2>&1 | tail -1)"
done; fi
for rep in $(seq $reps); do for v in "$@"; do
  cp $L/variants/$v.so $L/libmi355gs.so
  timeout 300 python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('%-10s bwd %.2f us  fwd %.2f us  | dropin %.0f  synced %.0f  run-ahead %.0f it/s' % ('$v', r['avg_kernel_ms']*1e3, r['composite_fwd']['avg_kernel_ms']*1e3, d['value'], d['loops']['one_call_synced'], d['loops']['one_call_run_ahead']))"
done; done
cp /tmp/keep.so $L/libmi355gs.so
