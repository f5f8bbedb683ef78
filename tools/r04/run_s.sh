#!/bin/bash
out=gpurun_out/r04s; mkdir -p $out; rm -f $out/summary.txt
old=superscreen_amd/lib/libssa_old.so
for i in 1 2 3; do
  for lib in $old superscreen_amd/lib/libsuperscreen_hip.so; do
    echo -n "$lib | " >> $out/summary.txt; SSA_LIB_PATH=$PWD/$lib timeout 300 python tools/fact_timing.py 2>&1 | tail -1 >> $out/summary.txt
  done
done
for lib in $old superscreen_amd/lib/libsuperscreen_hip.so; do
  echo -n "$lib | " >> $out/summary.txt; SSA_LIB_PATH=$PWD/$lib timeout 300 python tools/stack_timing.py 2>&1 | tail -1 >> $out/summary.txt
  echo -n "$lib | " >> $out/summary.txt; SSA_LIB_PATH=$PWD/$lib timeout 300 python tools/fact_timing.py auto float32 2>&1 | tail -1 >> $out/summary.txt
done
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k chol --timeout 300 > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/summary.txt
cat $out/summary.txt
