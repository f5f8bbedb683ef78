#!/bin/bash
out=gpurun_out/r04aa; mkdir -p $out; rm -f $out/summary.txt
for p in 0 1 0 1; do timeout 300 python tools/r04/pipelined_probe.py $p 2>&1 | tail -1 >> $out/summary.txt; done
cat $out/summary.txt
