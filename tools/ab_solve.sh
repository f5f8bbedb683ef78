#!/bin/bash
# A/B of library builds on the solve phase (same GPU box): bash tools/ab_solve.sh lib1.so lib2.so ...
for i in 1 2; do
  for lib in "$@"; do
    echo -n "$lib  "; SSA_LIB_PATH=$PWD/superscreen_amd/lib/$lib timeout 300 python tools/step_breakdown.py 2>&1 | grep "^factorize" | tail -1
  done
done
