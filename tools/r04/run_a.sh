#!/bin/bash
# round 4, first GPU call: correctness of the round / fold schedule, then factorization timings by switch
out=gpurun_out/r04a; mkdir -p $out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "chol" > $out/pytest_chol.log 2>&1; echo "pytest chol rc=$?" >> $out/summary.txt
tail -3 $out/pytest_chol.log >> $out/summary.txt
run() { echo -n "$1 | " >> $out/summary.txt; env $1 timeout 300 python tools/fact_timing.py ${2:-auto} ${3:-float64} ${4:-91} 2>&1 | tail -1 >> $out/summary.txt; }
for rep in 1 2; do
run "SSA_CHOL_TAIL=0 SSA_CHOL_FOLD=0"
run "SSA_CHOL_TAIL=0 SSA_CHOL_FOLD=1"
run "SSA_CHOL_TAIL=4096 SSA_CHOL_FOLD=0"
run "SSA_CHOL_TAIL=6144 SSA_CHOL_FOLD=0"
run "SSA_CHOL_TAIL=8192 SSA_CHOL_FOLD=0"
run "SSA_CHOL_TAIL=8192 SSA_CHOL_FOLD=1"
run "SSA_CHOL_TAIL=10240 SSA_CHOL_FOLD=1"
run "SSA_CHOL_TAIL=12288 SSA_CHOL_FOLD=1"
done
run "SSA_CHOL_TAIL=8192 SSA_CHOL_FOLD=1 SSA_CHOL_TAIL_EXCL=0"
run "SSA_CHOL_TAIL=8192 SSA_CHOL_FOLD=1 SSA_CHOL_TAIL_EXCL=100000"
run "SSA_CHOL_TAIL=8192 SSA_CHOL_FOLD=1 SSA_CHOL_TAIL_EXCL=300"
run "SSA_CHOL_TAIL=0 SSA_CHOL_FOLD=0" auto float32
run "SSA_CHOL_TAIL=8192 SSA_CHOL_FOLD=1" auto float32
cat $out/summary.txt
