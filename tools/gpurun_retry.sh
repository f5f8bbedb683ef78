#!/bin/bash
# gpurun with retries while every GPU slot of the pod is busy (status=transient: nothing charged).
#   bash tools/gpurun_retry.sh <timeout_s> '<command>'
for try in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2" > /tmp/gpurun_retry.$$ 2>&1
  if grep -q "status=transient" /tmp/gpurun_retry.$$; then sleep 90; else break; fi
done
cat /tmp/gpurun_retry.$$; rm -f /tmp/gpurun_retry.$$
