#!/bin/bash
out=gpurun_out/r04l; mkdir -p $out; rm -f $out/summary.txt
timeout 300 tools/probes/q_probe 91 > $out/q_probe_91.txt 2>&1
timeout 300 tools/probes/q_probe 129 > $out/q_probe_129.txt 2>&1
timeout 1200 python -X faulthandler -m pytest tests -q -m gpu --timeout 400 -k "float32_no_worse or helper_groups or chain_streams_are or profile_kinds or placement" > $out/pytest_new.log 2>&1; echo "pytest new rc=$?" >> $out/summary.txt; tail -3 $out/pytest_new.log >> $out/summary.txt
timeout 900 python -X faulthandler bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/summary.txt
cat $out/summary.txt
