#!/bin/bash
out=gpurun_out/r04m; mkdir -p $out
timeout 300 tools/probes/q_probe 91 > $out/q_probe_91.txt 2>&1
timeout 300 tools/probes/q_probe 129 > $out/q_probe_129.txt 2>&1
timeout 600 python tools/r04/f32_diag.py > $out/f32_diag.txt 2>&1
cat $out/f32_diag.txt; sed -n 1,20p $out/q_probe_91.txt; sed -n 1,20p $out/q_probe_129.txt
