#!/bin/bash
out=gpurun_out/r04q; mkdir -p $out; rm -f $out/band.txt
for b in 8 4 6 12 16 8; do SSA_SYRK_BAND=$b timeout 300 python tools/probes/syrk_band_probe.py 2>&1 | tail -1 >> $out/band.txt; done
cat $out/band.txt
