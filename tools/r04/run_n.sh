#!/bin/bash
out=gpurun_out/r04n; mkdir -p $out; rm -f $out/summary.txt
timeout 600 python tools/r04/f32_diag.py > $out/f32_diag.txt 2>&1
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu --timeout 400 > $out/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $out/summary.txt; tail -3 $out/pytest_gpu.log >> $out/summary.txt
timeout 900 python -X faulthandler bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/summary.txt
cat $out/summary.txt; grep -v amdgpu $out/f32_diag.txt
