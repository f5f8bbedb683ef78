#!/bin/bash
out=gpurun_out/r04i; mkdir -p $out
timeout 500 python -X faulthandler -m pytest tests/test_headline_gpu.py -x -v -m gpu -k two_film --timeout 150 > $out/two_film.log 2>&1; echo "rc=$?" >> $out/two_film.log
tail -60 $out/two_film.log
