#!/bin/bash
out=gpurun_out/r04g; mkdir -p $out
timeout 300 tools/probes/q_probe 91 > $out/q_probe_91.txt 2>&1
timeout 300 tools/probes/q_probe 129 > $out/q_probe_129.txt 2>&1
timeout 1800 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" > $out/summary.txt; tail -5 $out/pytest_gpu.log >> $out/summary.txt
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/summary.txt
cat $out/summary.txt; cat $out/q_probe_91.txt $out/q_probe_129.txt
