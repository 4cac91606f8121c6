#!/bin/bash
# usage (build container): tools/build_variant.sh <name> [extra hipcc flags...]
#   -> instantsplat_amd/lib/variants/<name>.so, the library built with the extra flags (A/B builds for tools/variants.sh and
#   tools/ab_bench.py on the GPU box; git-ignored, they travel with the gpurun snapshot).  Measurement helper, not product code.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; shift
tmp="$(mktemp -d)"
cp "$ROOT"/instantsplat_amd/csrc/*.hip "$ROOT"/instantsplat_amd/csrc/*.h "$ROOT"/instantsplat_amd/csrc/Makefile "$tmp"/
mkdir -p "$tmp/../include_stub" "$ROOT/instantsplat_amd/lib/variants"
# the sources include ../../include/mi355gs.h relative to csrc/: rebuild that layout around the temporary directory
work="$(mktemp -d)"; mkdir -p "$work/pkg/csrc" "$work/include" "$work/pkg/lib"
cp "$tmp"/* "$work/pkg/csrc/"; cp "$ROOT/include/mi355gs.h" "$work/include/"
make -C "$work/pkg/csrc" -j8 EXTRA="$*" ../lib/libmi355gs.so > "$work/build.log" 2>&1 || { tail -20 "$work/build.log"; exit 1; }
cp "$work/pkg/lib/libmi355gs.so" "$ROOT/instantsplat_amd/lib/variants/$name.so"
rm -rf "$tmp" "$work"
echo "built instantsplat_amd/lib/variants/$name.so ($*)"
