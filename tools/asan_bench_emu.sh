#!/bin/bash
# CPU only: bench.py's whole single-rank sequence (every loop of the line: the as-written loss through lazy_loss, the one-call
# trainer and its hook thread, run-ahead, the no-grad renders, the deterministic switch is not part of it) with BOTH native
# halves under AddressSanitizer — the kernel sources under the SIMT emulator (as tools/asan_emu.sh builds them) and
# csrc_torch/binding.cpp (the compiled autograd nodes, their caches and the pinned result words).  The fuzz of tools/asan_emu.sh
# goes through the ctypes binding and never enters binding.cpp; this does.
#   tools/asan_bench_emu.sh [pointmap edge] [res]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT=/tmp/mi355gs_asan; mkdir -p $OUT
for f in "$ROOT"/instantsplat_amd/csrc/*.hip; do
  g++ -x c++ -std=c++17 -O1 -g -fPIC -fsanitize=address -fno-omit-frame-pointer -I"$ROOT/tests/emu" -Wno-unused-function -Wno-attributes -ffp-contract=fast \
      -c "$f" -o $OUT/emu_$(basename "$f" .hip).o &
done; wait
g++ -shared -fsanitize=address -o $OUT/libmi355gs_emu_asan.so $OUT/emu_*.o
python - "$ROOT" "$OUT" <<'PY'
import os, subprocess, sys, sysconfig, torch
from torch.utils import cpp_extension as ce
root, out = sys.argv[1:3]
lib_dir = ce.library_paths()[0]
cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address", "-fno-omit-frame-pointer", "-Wno-unused-function",
       "-DTORCH_EXTENSION_NAME=_mi355gs_torch", "-DTORCH_API_INCLUDE_EXTENSION_H", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
       f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
for inc in ce.include_paths() + [sysconfig.get_paths()["include"], "/opt/rocm/include"]:
    cmd += ["-isystem", inc]
cmd += [os.path.join(root, "instantsplat_amd", "csrc_torch", "binding.cpp"), "-o", os.path.join(out, "_mi355gs_torch.so"), f"-L{lib_dir}",
        f"-Wl,-rpath,{lib_dir}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-ltorch_hip", "-ltorch_python"]
subprocess.check_call(cmd)
PY
cd /tmp
MI355GS_BENCH_CHILD=1 OMP_NUM_THREADS=1 ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 \
  LD_PRELOAD="$(gcc -print-file-name=libasan.so)" python - "$ROOT" "$OUT" "${1:-6}" "${2:-32}" <<'PY'
import runpy, sys
root, out, edge, res = sys.argv[1:5]
sys.path.insert(0, root)
from instantsplat_amd import _lib
_lib.EXT_PATH = out + "/_mi355gs_torch.so"          # the instrumented binding instead of lib/_mi355gs_torch.so
sys.argv = [root + "/bench.py", "--steps", "3", "--warmup", "1", "--pointmap", edge, "--res", res, "--cpu-iters", "0", "--attempt-seconds", "0",
            "--emulated-kernels", out + "/libmi355gs_emu_asan.so"]
runpy.run_path(root + "/bench.py", run_name="__main__")
print("bench.py under AddressSanitizer (emulated kernels + compiled binding): finished", file=sys.stderr)
PY
