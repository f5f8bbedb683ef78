#!/bin/bash
out=gpurun_out/r04w; mkdir -p $out; rm -f $out/summary.txt
for q in 8 4 16 2 32; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/r04/stack_pass_timing.py 100 4 2>&1 | tail -1 >> $out/summary.txt
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/r04/stack_pass_timing.py 91 2 2>&1 | tail -1 >> $out/summary.txt
done
timeout 300 python tools/r04/stack_pass_timing.py 100 4 2>&1 | tail -1 >> $out/summary.txt
cat $out/summary.txt
