#!/bin/bash
# A whole-library variant build: _diag/liblfsr_<tag>.so with extra compile flags on every source, e.g. tools/build_variant_all.sh nont "-DLFSR_NT_STORES=0"
set -e
cd "$(dirname "$0")/.."
P=$(ls -d ntire-2026-*_amd)/csrc
tag=$1; flags=$2
mkdir -p _diag/obj_$tag
for f in $(grep '^SRCS' $P/Makefile | sed 's/SRCS *:= *//'); do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize $flags -x hip -c $P/$f -o _diag/obj_$tag/$f.o &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 0.2; done
done
wait
OBJS=""; for f in $(grep '^SRCS' $P/Makefile | sed 's/SRCS *:= *//'); do OBJS="$OBJS _diag/obj_$tag/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -ldl -o _diag/liblfsr_$tag.so
rm -rf _diag/obj_$tag
ls -la _diag/liblfsr_$tag.so
