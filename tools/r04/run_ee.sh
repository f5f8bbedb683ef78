#!/bin/bash
out=gpurun_out/r04ee; mkdir -p $out; rm -f $out/summary.txt
for q in none 8 16; do
  if [ $q = none ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout 300 python tools/r04/stack_pass_timing.py 100 4 2>&1 | tail -1 >> $out/summary.txt
  timeout 300 python tools/r04/stack_pass_timing.py 100 3 2>&1 | tail -1 >> $out/summary.txt
done
unset GPU_MAX_HW_QUEUES
timeout 300 python tools/config5_timing.py 2>&1 | tail -1 >> $out/summary.txt
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/config5_timing.py 2>&1 | tail -1 >> $out/summary.txt
timeout 900 python -m pytest tests/test_solve_gpu.py tests/test_headline_gpu.py -x -q -m gpu --timeout 300 -k "stack or three or four or placement" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/summary.txt; tail -2 $out/pytest.log >> $out/summary.txt
cat $out/summary.txt
