#!/bin/bash
out=gpurun_out/r04vv; mkdir -p $out; rm -f $out/summary.txt
timeout 1200 python -X faulthandler tools/stress_factorization.py 100 > $out/stress.txt 2>&1; echo "stress rc=$?" >> $out/summary.txt; tail -4 $out/stress.txt >> $out/summary.txt
timeout 1800 python -X faulthandler -m pytest tests -q -m gpu --timeout 600 > $out/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $out/summary.txt; tail -1 $out/pytest_gpu.log >> $out/summary.txt
cat $out/summary.txt
