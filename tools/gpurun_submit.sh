#!/bin/bash
# submit a gpurun call, re-submitting while the pool answers "no slot free" (exit 3: nothing ran, nothing was charged)
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout ${2:-1200} -- "$1"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3
