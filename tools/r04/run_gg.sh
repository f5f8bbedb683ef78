#!/bin/bash
out=gpurun_out/r04gg; mkdir -p $out
timeout 1200 python -m pytest tests/test_headline_gpu.py -m gpu -x -q -s --timeout 1000 -k "configH_float32" > $out/pytest_f32.txt 2>&1
timeout 600 python tools/r04/f32_diag.py > $out/f32_diag.txt 2>&1
grep -n "config H float32\|passed\|failed\|Error" $out/pytest_f32.txt | head; cat $out/f32_diag.txt
