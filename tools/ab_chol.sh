#!/bin/bash
# A/B of two builds of the library on the same GPU box: bash tools/ab_chol.sh [method] [dtype] [K]
old=superscreen_amd/lib/libssa_old.so
new=superscreen_amd/lib/libsuperscreen_hip.so
for i in 1 2; do
  for lib in $old $new; do
    SSA_LIB_PATH=$PWD/$lib timeout 300 python tools/fact_timing.py "$@" 2>&1 | tail -1
  done
done
