#!/bin/bash
out=gpurun_out/r04j; mkdir -p $out
timeout 1500 python -X faulthandler -m pytest tests -x -q -m gpu --timeout 300 > $out/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" > $out/summary.txt; tail -5 $out/pytest_gpu.log >> $out/summary.txt
timeout 600 python -X faulthandler bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/summary.txt
cat $out/summary.txt; tail -3 $out/bench.err
