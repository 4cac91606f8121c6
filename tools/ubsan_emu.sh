#!/bin/bash
# CPU only: the kernel sources under the SIMT emulator and csrc_torch/binding.cpp built with UndefinedBehaviorSanitizer
# (-fsanitize=undefined -fno-sanitize-recover: shifts, signed overflow, misaligned / null accesses, out-of-range enum / bool loads,
# array bounds of the static LDS arrays), driven by the raster fuzz and the emulated-kernel test files.
# (float-cast-overflow is NOT enabled: float -> int of an out-of-range value is undefined in C++ and saturates on gfx950's
# v_cvt_i32_f32, which the kernels rely on for rectangles of Gaussians far outside the frame; the host compiler's behaviour
# there is what tests/emu has always run with.)
#   tools/ubsan_emu.sh [seed] [cases]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=/tmp/mi355gs_ubsan; mkdir -p $OUT
SAN="-fsanitize=undefined -fno-sanitize-recover=undefined -fno-sanitize=vptr"
for f in "$ROOT"/instantsplat_amd/csrc/*.hip; do
  g++ -x c++ -std=c++17 -O1 -g -fPIC $SAN -I"$ROOT/tests/emu" -Wno-unused-function -Wno-attributes -ffp-contract=fast \
      -c "$f" -o $OUT/emu_$(basename "$f" .hip).o &
done; wait
g++ -shared $SAN -o $OUT/libmi355gs_emu_ubsan.so $OUT/emu_*.o
python - "$ROOT" "$OUT" $SAN <<'PY'
import os, subprocess, sys, sysconfig, torch
from torch.utils import cpp_extension as ce
root, out = sys.argv[1:3]; san = sys.argv[3:]
lib_dir = ce.library_paths()[0]
cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function"] + san + [
       "-DTORCH_EXTENSION_NAME=_mi355gs_torch", "-DTORCH_API_INCLUDE_EXTENSION_H", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
       f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
for inc in ce.include_paths() + [sysconfig.get_paths()["include"], "/opt/rocm/include"]:
    cmd += ["-isystem", inc]
cmd += [os.path.join(root, "instantsplat_amd", "csrc_torch", "binding.cpp"), "-o", os.path.join(out, "_mi355gs_torch.so"), f"-L{lib_dir}",
        f"-Wl,-rpath,{lib_dir}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-ltorch_hip", "-ltorch_python"]
subprocess.check_call(cmd)
PY
cd "$ROOT"
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD="$(gcc -print-file-name=libubsan.so)"
MI355GS_EMU_LIB=$OUT/libmi355gs_emu_ubsan.so python tools/fuzz_raster_emu.py "${1:-0}" "${2:-40}"
MI355GS_EMU_LIB=$OUT/libmi355gs_emu_ubsan.so MI355GS_TORCH_EXT=$OUT/_mi355gs_torch.so OMP_NUM_THREADS=1 \
  python -m pytest -x -q -m "not gpu" -p no:cacheprovider tests/test_ops_emu.py tests/test_edge_emu.py tests/test_raster_emu.py tests/test_sora_emu.py tests/test_scene_io.py
