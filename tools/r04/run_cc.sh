#!/bin/bash
out=gpurun_out/r04cc; mkdir -p $out; rm -f $out/summary.txt
run() { echo -n "$1 | " >> $out/summary.txt; env $1 timeout 300 python $2 $3 $4 $5 2>&1 | tail -1 >> $out/summary.txt; }
for rep in 1 2; do
for d in 2 3 4; do run "SSA_CHOL_DEPTH=$d" tools/fact_timing.py; done
done
for d in 2 3 4; do run "SSA_CHOL_DEPTH=$d" tools/fact_single.py 129; run "SSA_CHOL_DEPTH=$d" tools/stack_timing.py; run "SSA_CHOL_DEPTH=$d" tools/fact_timing.py auto float32; done
SSA_CHOL_DEPTH=3 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k chol --timeout 300 > $out/pytest.log 2>&1; echo "pytest depth3 rc=$?" >> $out/summary.txt
cat $out/summary.txt
