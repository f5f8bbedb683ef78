#!/bin/bash
out=gpurun_out/r04c; mkdir -p $out; rm -f $out/summary.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "chol" > $out/pytest_chol.log 2>&1; echo "pytest chol rc=$?" >> $out/summary.txt
tail -3 $out/pytest_chol.log >> $out/summary.txt
run() { echo -n "$1 | " >> $out/summary.txt; env $1 timeout 300 python tools/fact_timing.py ${2:-auto} ${3:-float64} ${4:-91} 2>&1 | tail -1 >> $out/summary.txt; }
for rep in 1 2; do
run "SSA_CHOL_TAIL=0"
run "SSA_CHOL_TAIL=8192"
run "SSA_CHOL_TAIL=10240"
run "SSA_CHOL_TAIL=12288"
run "SSA_CHOL_TAIL=10240 SSA_CHOL_FILL_TILES=800"
run "SSA_CHOL_TAIL=10240 SSA_CHOL_FILL_TILES=3000"
run "SSA_CHOL_TAIL=10240 SSA_CHOL_FILL_TILES=100000"
run "SSA_CHOL_TAIL=10240 SSA_CHOL_TAIL_EXCL=0"
run "SSA_CHOL_TAIL=10240 SSA_CHOL_TAIL_EXCL=2500"
done
cd /tmp; export TMPDIR=/tmp
SSA_CHOL_TAIL=10240 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/rt -- python3 $GRAFT_REPO_ROOT/tools/r04/round_timeline.py run float64 > $GRAFT_REPO_ROOT/$out/rt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/r04/round_timeline.py analyse $out/rt > $out/rt_timeline.txt 2>&1
rm -rf $out/rt
cat $out/summary.txt
