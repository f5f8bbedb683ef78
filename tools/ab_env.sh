#!/bin/bash
# A/B of an environment switch on the same GPU box: bash tools/ab_env.sh VAR=a VAR=b -- [method] [dtype] [K]
a=$1; b=$2; shift 3
for i in 1 2; do
  for kv in $a $b; do
    echo -n "$kv  "; env $kv timeout 300 python tools/fact_timing.py "$@" 2>&1 | tail -1
  done
done
