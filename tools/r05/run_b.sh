#!/bin/bash
# round 5, call B: which part of the four-film schedule races?  The plain schedule, then variants (chol.hip: CholDebug).
out=gpurun_out/r05b; mkdir -p $out; rm -f $out/summary.txt
run() {  # name, reps, env
  SSA_CHOL_DEBUG="$3" timeout 600 python -X faulthandler tools/chol_race_hunt.py $2 > $out/hunt_$1.txt 2>&1
  echo "$1 [$3] rc=$? $(tail -1 $out/hunt_$1.txt)" >> $out/summary.txt
}
run plain 150 ""
run nosplit 100 "split=0"
run norounds 100 "tail=0"
run late 100 "late=1"
run nodelay 100 "delay=0"
run plain2 150 ""
cat $out/summary.txt
grep -h -A2 DIFFERENT $out/hunt_*.txt | head -60
