#!/bin/bash
# Variants of wgrad.hip for timing: _diag/liblfsr_wg_<tag>.so for each "tag:flags" argument (reuses csrc/build/*.o)
set -e
cd "$(dirname "$0")/.."
P=$(ls -d ntire-2026-*_amd)/csrc
mkdir -p _diag/obj
for a in "$@"; do
  tag=${a%%:*}; flags=${a#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize $flags -x hip -c $P/wgrad.hip -o _diag/obj/wg_$tag.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $P/build/*.o | grep -v "wgrad.hip.o") _diag/obj/wg_$tag.o -ldl -o _diag/liblfsr_wg_$tag.so ) &
done
wait
ls _diag/liblfsr_wg_*.so
