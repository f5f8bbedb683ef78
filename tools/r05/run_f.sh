#!/bin/bash
# round 5, call F: Q assembly as one-shot pieces against the strips
out=gpurun_out/r05f; mkdir -p $out; rm -f $out/summary.txt
timeout 600 python -X faulthandler -m pytest tests -q -m gpu --timeout 600 -k "q_assemble or Q_property or single_film_vs_reference or c_abi_error or tiny_and_degenerate" > $out/pytest_subset.log 2>&1; echo "pytest subset rc=$?" >> $out/summary.txt; tail -2 $out/pytest_subset.log >> $out/summary.txt
timeout 900 python -X faulthandler tools/q_form_timing.py 91 129 > $out/q_form_timing.txt 2>&1; echo "q_form_timing rc=$?" >> $out/summary.txt
cat $out/summary.txt; cat $out/q_form_timing.txt
