#!/bin/bash
out=gpurun_out/r04f; mkdir -p $out; rm -f $out/summary.txt
runp() { echo -n "$1 | $2 $3 $4 | " >> $out/summary.txt; env $1 timeout 600 python $2 $3 $4 $5 2>&1 | tail -1 >> $out/summary.txt; }
for rep in 1 2; do
for T in 0 10240 20480 60000; do runp "SSA_CHOL_TAIL=$T" tools/fact_single.py 129; done
for T in 0 10240 16384 24576 60000; do runp "SSA_CHOL_TAIL=$T" tools/stack_timing.py; done
for T in 0 8192 10240 14336; do runp "SSA_CHOL_TAIL=$T" tools/fact_timing.py auto float32; done
for T in 0 6144 10240 20000; do runp "SSA_CHOL_TAIL=$T" tools/fact_timing.py auto float64 81; done
for T in 0 10240 60000; do runp "SSA_CHOL_TAIL=$T" tools/fact_single.py 64; done
done
cat $out/summary.txt
