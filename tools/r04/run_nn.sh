#!/bin/bash
out=gpurun_out/r04nn; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q -s --timeout 900 -k "float32 or two_film_vs_oracle or sweep_64 or lu" > $out/pytest.txt 2>&1
grep -n "config H float32\|passed\|failed\|Error\|assert" $out/pytest.txt | head -20
