#!/bin/bash
# round 5, call C: the rounds' diagonal blocks as their workgroups read them (trace), and without exclusive launches
out=gpurun_out/r05c; mkdir -p $out; rm -f $out/summary.txt
run() {  # name, reps, env
  SSA_CHOL_DEBUG="$3" timeout 900 python -X faulthandler tools/chol_race_hunt.py $2 > $out/hunt_$1.txt 2>&1
  echo "$1 [$3] rc=$? $(tail -1 $out/hunt_$1.txt)" >> $out/summary.txt
}
run trace 400 "trace=1"
run noexcl 200 "excl=0"
run late 200 "late=1"
cat $out/summary.txt
grep -h -A8 DIFFERENT $out/hunt_*.txt | head -120
