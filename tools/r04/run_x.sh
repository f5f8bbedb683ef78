#!/bin/bash
out=gpurun_out/r04x; mkdir -p $out; rm -f $out/summary.txt
for q in 8 16 4; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/r04/queue_probe.py 4 2>&1 | tail -1 >> $out/summary.txt
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/r04/queue_probe.py 4 nofact 2>&1 | tail -1 >> $out/summary.txt
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/r04/queue_probe.py 2 2>&1 | tail -1 >> $out/summary.txt
done
cat $out/summary.txt
