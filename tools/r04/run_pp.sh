#!/bin/bash
out=gpurun_out/r04pp; mkdir -p $out; rm -f $out/summary.txt
timeout 900 python -X faulthandler tools/stress_factorization.py 40 > $out/stress.txt 2>&1; echo "stress rc=$?" >> $out/summary.txt; tail -4 $out/stress.txt >> $out/summary.txt
timeout 2400 python tools/collect_profiles.py r04 $out > $out/collect.log 2>&1; echo "collect rc=$?" >> $out/summary.txt
cat $out/summary.txt
