#!/bin/bash
# Ablation / parameter variants of the F(4x4,3x3) conv kernel for timing: _diag/liblfsr_w4_<tag>.so for each "tag:flags" argument,
# e.g. tools/build_w4_abl.sh "a1:-DW4_ABL=1" "r8:-DW4_URING=8".  Needs the product build (reuses csrc/build/*.o).
set -e
cd "$(dirname "$0")/.."
P=$(ls -d ntire-2026-*_amd)/csrc
mkdir -p _diag/obj
for a in "$@"; do
  tag=${a%%:*}; flags=${a#*:}
  EXTRA=""; SKIP="conv3x3_wino4.hip.o"
  case "$flags" in *LFSR_CONV_DIAG*)   # the stamp-buffer global lives in index_ops.hip of a DIAG build
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -DLFSR_CONV_DIAG -x hip -c $P/index_ops.hip -o _diag/obj/index_ops_diag.o
    EXTRA="_diag/obj/index_ops_diag.o"; SKIP="conv3x3_wino4.hip.o\|index_ops";;
  esac
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize $flags -x hip -c $P/conv3x3_wino4.hip -o _diag/obj/w4_$tag.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $P/build/*.o | grep -v "$SKIP") $EXTRA _diag/obj/w4_$tag.o -o _diag/liblfsr_w4_$tag.so ) &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.2; done
done
wait
ls _diag/liblfsr_w4_*.so
