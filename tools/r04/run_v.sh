#!/bin/bash
out=gpurun_out/r04v; mkdir -p $out; rm -f $out/summary.txt
for q in 8 4 16 8 4 16; do
  echo -n "GPU_MAX_HW_QUEUES=$q | " >> $out/summary.txt; GPU_MAX_HW_QUEUES=$q timeout 300 python tools/config5_timing.py 2>&1 | tail -1 >> $out/summary.txt
  echo -n "GPU_MAX_HW_QUEUES=$q | " >> $out/summary.txt; GPU_MAX_HW_QUEUES=$q timeout 300 python tools/fact_timing.py 2>&1 | tail -1 >> $out/summary.txt
done
cat $out/summary.txt
