#!/bin/bash
out=gpurun_out/r04o; mkdir -p $out; rm -f $out/summary.txt
timeout 900 python -X faulthandler -m pytest tests -q -m gpu --timeout 400 -k "assemble or system or q_matrix or Q_ or disk or golden" > $out/pytest_asm.log 2>&1; echo "pytest asm rc=$?" >> $out/summary.txt; tail -3 $out/pytest_asm.log >> $out/summary.txt
python tools/baseline_configs.py > $out/baseline_configs.txt 2>&1 || true
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pt -- python3 $GRAFT_REPO_ROOT/tools/pass_trace.py > $GRAFT_REPO_ROOT/$out/pt.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/rt -- python3 $GRAFT_REPO_ROOT/tools/r04/round_timeline.py run float64 > $GRAFT_REPO_ROOT/$out/rt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pass_trace.py analyse $out/pt > $out/pass_trace.txt 2>&1
python tools/r04/round_timeline.py analyse $out/rt > $out/rt_timeline.txt 2>&1
rm -rf $out/pt $out/rt
cat $out/summary.txt $out/pass_trace.txt
