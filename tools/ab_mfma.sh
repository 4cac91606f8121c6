#!/bin/bash
# GPU box: A/B of the backward's cross-row reduction on the matrix pipe (GS_BW_MFMA build) against the shipped build.
cd "$GRAFT_REPO_ROOT"
L=instantsplat_amd/lib
tools/ubench/mfma_mix > gpurun_out/r03_ubench_mfma_mix.txt 2>&1; tail -30 gpurun_out/r03_ubench_mfma_mix.txt
cp $L/libmi355gs.so /tmp/keep.so
for v in base bwd_mfma base bwd_mfma; do
  cp $L/variants/$v.so $L/libmi355gs.so
  echo "== $v"
  bash tools/trace_seq.sh | grep -E "fwd|bwd" | grep -v "false"
done > gpurun_out/r03_ab_bwd_mfma_kernel_avg.txt 2>&1
cat gpurun_out/r03_ab_bwd_mfma_kernel_avg.txt
for v in base bwd_mfma; do
  cp $L/variants/$v.so $L/libmi355gs.so
  bash tools/pmc.sh ab_$v SQ_INSTS_VALU,SQ_INSTS_MFMA,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES,SQ_ACTIVE_INST_VALU python bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-long-run | grep -E "kernel|composite"
done
cp $L/variants/bwd_mfma.so $L/libmi355gs.so
timeout 400 python -m pytest tests/test_raster_gpu.py tests/test_edge_gpu.py tests/test_properties_gpu.py -x -q 2>&1 | tail -3
cp /tmp/keep.so $L/libmi355gs.so
