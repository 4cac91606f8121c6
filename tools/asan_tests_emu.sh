#!/bin/bash
# (libstdc++ is preloaded next to libasan so that its __cxa_throw interceptor resolves: the tests expect C++ exceptions.)
# CPU only: the emulated-kernel test files of the CPU tier under AddressSanitizer — instrumented kernel sources (SIMT emulator)
# AND instrumented csrc_torch/binding.cpp, as tools/asan_bench_emu.sh builds them (run that first: it leaves both in /tmp/mi355gs_asan).
#   tools/asan_tests_emu.sh [pytest arguments; default: the *_emu.py files]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=/tmp/mi355gs_asan
test -f $OUT/libmi355gs_emu_asan.so -a -f $OUT/_mi355gs_torch.so || { echo "run tools/asan_bench_emu.sh first"; exit 2; }
cd "$ROOT"
ARGS=("$@"); [ ${#ARGS[@]} -eq 0 ] && ARGS=(tests/test_ops_emu.py tests/test_edge_emu.py tests/test_raster_emu.py tests/test_sora_emu.py tests/test_scene_io.py)
MI355GS_EMU_LIB=$OUT/libmi355gs_emu_asan.so MI355GS_TORCH_EXT=$OUT/_mi355gs_torch.so OMP_NUM_THREADS=1 \
  ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)" \
  python -m pytest -x -q -m "not gpu" -p no:cacheprovider "${ARGS[@]}"
