#!/bin/bash
# round 4, call 29: what a half tile (18 of 36 position products, 5 of 9 fragments per stage) would cost: timing ablation of the whole launch in that mode
set -e
mkdir -p gpurun_out/r4
P=$(ls -d ntire-2026-*_amd)
AB_ROUNDS=5 AB_GEOMS="800:n,200:n,25:n" timeout -k 10 300 python tools/conv_ab.py clk=_diag/liblfsr_w4_clk.so halfclk=_diag/liblfsr_w4_halfclk.so halfnt=_diag/liblfsr_w4_halfnt.so halfnp=_diag/liblfsr_w4_halfnp.so > gpurun_out/r4/c29b_ab.log 2>&1 || { tail -20 gpurun_out/r4/c29b_ab.log; exit 1; }
grep -v "^check" gpurun_out/r4/c29b_ab.log | grep -v "n_img=25"
