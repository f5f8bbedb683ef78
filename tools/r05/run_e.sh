#!/bin/bash
# round 5, call E: new tests; config-2 switch point of the rounds (single film); non-temporal loads in the solves' GEMVs
out=gpurun_out/r05e; mkdir -p $out; rm -f $out/summary.txt
timeout 900 python -X faulthandler -m pytest tests -q -m gpu --timeout 600 -k "sliced_or_whole or configH_full_size_vs_oracle or config5" > $out/pytest_subset.log 2>&1; echo "pytest subset rc=$?" >> $out/summary.txt; tail -2 $out/pytest_subset.log >> $out/summary.txt
for tail in "" "tail=0" "tail=4096" "tail=16384" "tail=24576" "tail=45000" ""; do
  SSA_CHOL_DEBUG="$tail" timeout 300 python tools/fact_single.py 129 2>/dev/null | tail -1 | sed "s/^/[$tail] /" >> $out/summary.txt
done
for rep in 1 2; do
  timeout 300 python tools/warm_solve_timing.py 2>/dev/null | tail -1 >> $out/summary.txt
  SSA_LIB_PATH=$PWD/superscreen_amd/lib/libssa_gemvnt.so timeout 300 python tools/warm_solve_timing.py 2>/dev/null | tail -1 >> $out/summary.txt
done
cat $out/summary.txt
