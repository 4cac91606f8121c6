#!/bin/bash
# CPU only: the kernel sources under the SIMT emulator, built with AddressSanitizer, driven by the raster fuzz (tools/fuzz_raster_emu.py).
# Catches out-of-bounds accesses of global buffers and of the kernels' LDS arrays (static arrays under the emulator).
#   tools/asan_emu.sh <seed> <cases>
# (ASan warns that it does not fully support the emulator's swapcontext fibers; detect_stack_use_after_return is off for that reason.)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=/tmp/mi355gs_asan; mkdir -p $OUT
for f in "$ROOT"/instantsplat_amd/csrc/*.hip; do
  g++ -x c++ -std=c++17 -O1 -g -fPIC -fsanitize=address -fno-omit-frame-pointer -I"$ROOT/tests/emu" -Wno-unused-function -Wno-attributes -ffp-contract=fast \
      -c "$f" -o $OUT/emu_$(basename "$f" .hip).o &
done; wait
g++ -shared -fsanitize=address -o $OUT/libmi355gs_emu_asan.so $OUT/emu_*.o
cd /tmp
MI355GS_EMU_LIB=$OUT/libmi355gs_emu_asan.so MI355GS_BINDING=ctypes ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 \
  LD_PRELOAD="$(gcc -print-file-name=libasan.so)" python "$ROOT/tools/fuzz_raster_emu.py" "${1:-0}" "${2:-40}"
