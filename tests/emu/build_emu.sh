#!/bin/bash
# TEST INFRASTRUCTURE: compile the unmodified kernel sources against the SIMT emulator header
# (tests/emu/hip/hip_runtime.h) with plain g++ -> tests/emu/libmi355gs_emu.so (CPU tests only).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../../instantsplat_amd/csrc"
OUT="$HERE/libmi355gs_emu.so"
newest=$(ls -t "$SRC"/*.hip "$SRC"/*.h "$HERE"/hip/hip_runtime.h "$HERE/../../include/mi355gs.h" | head -1)
if [ -f "$OUT" ] && [ "$OUT" -nt "$newest" ]; then exit 0; fi
objs=""
pids=""
for f in "$SRC"/*.hip; do
  o="$HERE/emu_$(basename "$f" .hip).o"
  rm -f "$o"   # a failed compile must not leave an older object to be linked
  g++ -x c++ -std=c++17 -O2 -g -fPIC -I"$HERE" -Wno-unused-function -Wno-attributes -ffp-contract=fast -c "$f" -o "$o" &
  pids="$pids $!"
  objs="$objs $o"
done
for p in $pids; do wait "$p"; done   # set -e: any failed compile fails the build (a bare `wait` would hide it)
rm -f "$OUT"
g++ -shared -o "$OUT" $objs
