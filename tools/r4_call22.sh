#!/bin/bash
# round 4, call 22: Winograd weight gradient: contiguous tile ranges per block against every gridDim-th tile (two processes, alternating)
mkdir -p gpurun_out/r4
for rep in 1 2 3; do
  python tools/wgrad_time.py 2>/dev/null | grep -v amdgpu
  LFSR_HIP_LIB=$PWD/_diag/liblfsr_wg_contig.so python tools/wgrad_time.py 2>/dev/null | grep -v amdgpu
done | tee gpurun_out/r4/c22_wgrad_contig.log
