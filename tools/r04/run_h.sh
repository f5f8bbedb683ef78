#!/bin/bash
out=gpurun_out/r04h; mkdir -p $out
for w in lu chol chol_then_lu; do
  echo "== $w" >> $out/hang.txt
  timeout 120 python tools/r04/hang_probe.py $w 64 >> $out/hang.txt 2>&1; echo "rc=$?" >> $out/hang.txt
done
echo "== bench small" >> $out/hang.txt
timeout 200 python -X faulthandler bench.py --steps 2 --warmup 1 --no-extras >> $out/hang.txt 2>&1; echo "rc=$?" >> $out/hang.txt
cat $out/hang.txt
