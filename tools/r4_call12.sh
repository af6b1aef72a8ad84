#!/bin/bash
# round 4, call 12: EPI attention with LDS layouts that are conflict-free for the hardware's lane groups: A/B against the round-3 layouts, the attention tests
set -e
mkdir -p gpurun_out/r4
P=$(ls -d ntire-2026-*_amd)
timeout -k 10 300 python tools/attn_ab.py old=_diag/liblfsr_attn_old.so new=$P/liblfsr_hip.so hb2=_diag/liblfsr_attn_mfma_hb2.so hb1=_diag/liblfsr_attn_mfma_hb1.so > gpurun_out/r4/c12_attn_ab.log 2>&1 || { tail -20 gpurun_out/r4/c12_attn_ab.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r4/c12_attn_ab.log


