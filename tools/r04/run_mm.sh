#!/bin/bash
out=gpurun_out/r04mm; mkdir -p $out
timeout 800 python tools/r04/f32_lu_emulation.py 91 > $out/f32_lu.txt 2>&1
timeout 600 python tools/r04/f32_diag.py > $out/f32_diag.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 -k "lu or float32" > $out/pytest_lu.txt 2>&1
tail -12 $out/f32_lu.txt; cat $out/f32_diag.txt; tail -4 $out/pytest_lu.txt
