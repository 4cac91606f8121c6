#!/bin/bash
# usage (GPU box): tools/variants.sh name...   -> per-kernel averages for each prebuilt instantsplat_amd/lib/variants/<name>.so
cd "$GRAFT_REPO_ROOT"
L=instantsplat_amd/lib
cp $L/libmi355gs.so /tmp/keep.so
for v in "$@"; do
  cp $L/variants/$v.so $L/libmi355gs.so
  echo "== $v"
  bash tools/trace_seq.sh | grep -E "${GREP:-count|scatter|sort|fwd|bwd}" | grep -v "false"
done
cp /tmp/keep.so $L/libmi355gs.so
