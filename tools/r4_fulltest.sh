#!/bin/bash
# the whole GPU suite, log under gpurun_out/r4
mkdir -p gpurun_out/r4
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4/${1:-fulltest}.log 2>&1; rc=$?
tail -5 gpurun_out/r4/${1:-fulltest}.log
exit $rc
