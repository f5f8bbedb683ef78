#!/bin/bash
# round 5, call A: where does the config-5 difference come from?  (1) the repeat loop as it is, (2) with poisoned
# buffers, (3) the whole GPU suite without -x, (4) the suite's cheap part under poison.
out=gpurun_out/r05a; mkdir -p $out; rm -f $out/summary.txt
timeout 900 python -X faulthandler tools/config5_repeat.py 8 > $out/repeat_plain.txt 2>&1; echo "repeat plain rc=$?" >> $out/summary.txt
SSA_POISON=big timeout 900 python -X faulthandler tools/config5_repeat.py 3 > $out/repeat_big.txt 2>&1; echo "repeat big rc=$?" >> $out/summary.txt
SSA_POISON=nan timeout 900 python -X faulthandler tools/config5_repeat.py 3 > $out/repeat_nan.txt 2>&1; echo "repeat nan rc=$?" >> $out/summary.txt
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu --timeout 600 > $out/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $out/summary.txt; tail -3 $out/pytest_gpu.log >> $out/summary.txt
SSA_POISON=big timeout 1500 python -X faulthandler -m pytest tests -q -m gpu --timeout 600 > $out/pytest_gpu_big.log 2>&1; echo "pytest big rc=$?" >> $out/summary.txt; tail -3 $out/pytest_gpu_big.log >> $out/summary.txt
cat $out/summary.txt; tail -30 $out/repeat_plain.txt; tail -12 $out/repeat_big.txt; tail -12 $out/repeat_nan.txt
