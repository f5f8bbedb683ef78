#!/bin/bash
# A/B of library builds on the cold step of config H (same GPU box): bash tools/ab_lib.sh lib1.so lib2.so ...
for i in 1 2; do
  for lib in "$@"; do
    echo -n "$lib  "; SSA_LIB_PATH=$PWD/superscreen_amd/lib/$lib timeout 600 python tools/ab_cold_step.py 91 10 8 2>&1 | grep "expected_passes=11"
  done
done
