#!/bin/bash
# Ablation variants of the Winograd conv kernel for timing (results are wrong by construction): _diag/liblfsr_diag_abl<N>.so
# needs tools/build_diag.sh first (reuses its objects)
set -e
cd "$(dirname "$0")/.."
P=$(ls -d ntire-2026-*_amd)/csrc
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -DLFSR_CONV_DIAG -DWINO_ABL=$n -x hip -c $P/conv3x3_wino.hip -o _diag/obj/wino_abl$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls _diag/obj/*.o | grep -v "conv3x3_wino.hip.o\|wino_abl") _diag/obj/wino_abl$n.o -o _diag/liblfsr_diag_abl$n.so
done
ls -la _diag/*.so
