#!/bin/bash
# round 3, call 1: gradient-parity attribution (tools/grad_parity.py) + baseline bench lines on this box + training step time per dgrad selection
set -e
mkdir -p gpurun_out/r3
python tools/grad_parity.py gpurun_out/r3/grad_parity.json > gpurun_out/r3/grad_parity.log 2>&1
python bench.py --no-cpu-baseline > gpurun_out/r3/c1_bench.json 2> gpurun_out/r3/c1_bench.err
python bench.py --workload train --steps 10 > gpurun_out/r3/c1_train_default.json 2>> gpurun_out/r3/c1_bench.err
LFSR_DGRAD3=wino2 python bench.py --workload train --steps 10 > gpurun_out/r3/c1_train_dgrad_wino2.json 2>> gpurun_out/r3/c1_bench.err
LFSR_DGRAD3=halo python bench.py --workload train --steps 10 > gpurun_out/r3/c1_train_dgrad_halo.json 2>> gpurun_out/r3/c1_bench.err
LFSR_CONV3X3=wino2 python bench.py --workload train --steps 10 > gpurun_out/r3/c1_train_wino2.json 2>> gpurun_out/r3/c1_bench.err
tail -30 gpurun_out/r3/grad_parity.log
cat gpurun_out/r3/c1_*.json | cut -c1-400
