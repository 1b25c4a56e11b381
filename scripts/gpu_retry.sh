#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged)
# usage: gpu_retry.sh TIMEOUT 'command'
for i in $(seq 1 20); do
  gpurun --timeout "$1" -- "$2"; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
