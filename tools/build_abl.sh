#!/bin/bash
# Ablation / parameter variants of ONE translation unit for timing: _diag/liblfsr_<tag>.so for each "tag:flags" argument, e.g.
#   tools/build_abl.sh epi_b3.hip "a1:-DEB_ABL=1" "a16:-DEB_ABL=16"
# Needs the product build (reuses csrc/build/*.o).  The variants are loaded with LFSR_HIP_LIB=<path> (capi.py).
set -e
cd "$(dirname "$0")/.."
P=$(ls -d ntire-2026-*_amd)/csrc
src=$1; shift
mkdir -p _diag/obj
for a in "$@"; do
  tag=${a%%:*}; flags=${a#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize $flags -x hip -c $P/$src -o _diag/obj/${src%.*}_$tag.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $P/build/*.o | grep -v "/$src.o") _diag/obj/${src%.*}_$tag.o -ldl -o _diag/liblfsr_${src%.*}_$tag.so ) &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.2; done
done
wait
ls _diag/liblfsr_${src%.*}_*.so
