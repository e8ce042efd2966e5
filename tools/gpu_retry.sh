#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3: nothing charged).  usage: tools/gpu_retry.sh <timeout s> '<command>'
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
