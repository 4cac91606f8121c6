#!/bin/bash
# GPU box: tools/ab_variants.sh <variant>...   per-kernel averages (alternating twice), then parity tests on the LAST variant
cd "$GRAFT_REPO_ROOT"
L=instantsplat_amd/lib
cp $L/libmi355gs.so /tmp/keep.so
for rep in 1 2; do for v in "$@"; do
  cp $L/variants/$v.so $L/libmi355gs.so
  echo "== $v"
  bash tools/trace_seq.sh | grep -E "${GREP:-fwd|bwd}" | grep -v "true>"
done; done > gpurun_out/ab_kernel_avg.txt 2>&1
cat gpurun_out/ab_kernel_avg.txt
last="${@: -1}"
cp $L/variants/$last.so $L/libmi355gs.so
bash tools/pmc.sh ab_$last SQ_INSTS_VALU,SQ_INSTS_LDS,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run | grep -E "kernel|composite"
timeout 600 python -m pytest ${TESTS:-tests/test_raster_gpu.py tests/test_edge_gpu.py tests/test_properties_gpu.py tests/test_baseline_sizes_gpu.py} -x -q 2>&1 | tail -3
cp /tmp/keep.so $L/libmi355gs.so
