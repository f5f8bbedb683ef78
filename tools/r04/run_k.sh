#!/bin/bash
out=gpurun_out/r04k; mkdir -p $out; rm -f $out/summary.txt
runp() { echo -n "$1 | $2 $3 $4 | " >> $out/summary.txt; env $1 timeout 600 python $2 $3 $4 $5 2>&1 | tail -1 >> $out/summary.txt; }
runp "A=1" tools/fact_timing.py
runp "A=1" tools/fact_timing.py auto float32
runp "A=1" tools/fact_single.py 129
runp "A=1" tools/stack_timing.py
runp "A=1" tools/fact_timing.py
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu --timeout 300 > $out/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $out/summary.txt; tail -3 $out/pytest_gpu.log >> $out/summary.txt
timeout 600 python -X faulthandler bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/summary.txt
cat $out/summary.txt
