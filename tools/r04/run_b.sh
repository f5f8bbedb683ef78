#!/bin/bash
out=gpurun_out/r04b; mkdir -p $out; rm -f $out/summary.txt
run() { echo -n "$1 | " >> $out/summary.txt; env $1 timeout 300 python tools/fact_timing.py ${2:-auto} ${3:-float64} ${4:-91} 2>&1 | tail -1 >> $out/summary.txt; }
for rep in 1 2; do
run "SSA_CHOL_TAIL=0 SSA_CHOL_FOLD=0"
run "SSA_CHOL_TAIL=8192 SSA_CHOL_FOLD=0"
run "SSA_CHOL_TAIL=10240 SSA_CHOL_FOLD=0"
run "SSA_CHOL_TAIL=12288 SSA_CHOL_FOLD=0"
run "SSA_CHOL_TAIL=16384 SSA_CHOL_FOLD=0"
run "SSA_CHOL_TAIL=30000 SSA_CHOL_FOLD=0"
run "SSA_CHOL_TAIL=0 SSA_CHOL_FOLD=1"
run "SSA_CHOL_TAIL=8192 SSA_CHOL_FOLD=1"
run "SSA_CHOL_TAIL=12288 SSA_CHOL_FOLD=1"
done
cd /tmp; export TMPDIR=/tmp
SSA_CHOL_TAIL=8192 SSA_CHOL_FOLD=0 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/rt8k -- python3 $GRAFT_REPO_ROOT/tools/r04/round_timeline.py run float64 > $GRAFT_REPO_ROOT/$out/rt8k.log 2>&1
SSA_CHOL_TAIL=30000 SSA_CHOL_FOLD=0 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/rtall -- python3 $GRAFT_REPO_ROOT/tools/r04/round_timeline.py run float64 > $GRAFT_REPO_ROOT/$out/rtall.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/r04/round_timeline.py analyse $out/rt8k > $out/rt8k_timeline.txt 2>&1
python tools/r04/round_timeline.py analyse $out/rtall > $out/rtall_timeline.txt 2>&1
rm -rf $out/rt8k $out/rtall
timeout 300 tools/probes/q_probe 91 > $out/q_probe.txt 2>&1
cat $out/summary.txt
