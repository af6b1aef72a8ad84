#!/bin/bash
# Diagnostic build of the library (in-kernel s_memtime stamps in the conv kernels): _diag/liblfsr_diag.so
set -e
cd "$(dirname "$0")/.."
P=$(ls -d ntire-2026-*_amd)/csrc
mkdir -p _diag/obj
for f in $(grep '^SRCS' $P/Makefile | sed 's/SRCS *:= *//'); do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-slp-vectorize -DLFSR_CONV_DIAG -x hip -c $P/$f -o _diag/obj/$f.o &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 0.2; done
done
wait
OBJS=""; for f in $(grep '^SRCS' $P/Makefile | sed 's/SRCS *:= *//'); do OBJS="$OBJS _diag/obj/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o _diag/liblfsr_diag.so
ls -la _diag/liblfsr_diag.so
