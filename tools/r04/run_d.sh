#!/bin/bash
out=gpurun_out/r04d; mkdir -p $out; rm -f $out/summary.txt
run() { echo -n "$1 | " >> $out/summary.txt; env $1 timeout 300 python tools/fact_timing.py ${2:-auto} ${3:-float64} ${4:-91} 2>&1 | tail -1 >> $out/summary.txt; }
export SSA_CHOL_TAIL=10240
for rep in 1 2; do
run "SSA_CHOL_SLICE_WGS=256"
run "SSA_CHOL_SLICE_WGS=128"
run "SSA_CHOL_SLICE_WGS=64"
run "SSA_CHOL_SLICE_WGS=128 SSA_CHOL_SLICE_K=256"
run "SSA_CHOL_SLICE_WGS=128 SSA_CHOL_LATE_TRANSPOSE=1"
run "SSA_CHOL_SLICE_WGS=64 SSA_CHOL_LATE_TRANSPOSE=1"
run "SSA_CHOL_SLICE_WGS=256 SSA_CHOL_LATE_TRANSPOSE=1"
run "SSA_CHOL_SLICE_WGS=96 SSA_CHOL_LATE_TRANSPOSE=1 SSA_CHOL_TAIL_EXCL=1500"
run "SSA_CHOL_FILL_TILES=0"
done
cd /tmp; export TMPDIR=/tmp
SSA_CHOL_SLICE_WGS=128 SSA_CHOL_LATE_TRANSPOSE=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/rt -- python3 $GRAFT_REPO_ROOT/tools/r04/round_timeline.py run float64 > $GRAFT_REPO_ROOT/$out/rt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/r04/round_timeline.py analyse $out/rt > $out/rt_timeline.txt 2>&1
rm -rf $out/rt
cat $out/summary.txt
