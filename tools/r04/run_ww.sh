#!/bin/bash
out=gpurun_out/r04ww; mkdir -p $out; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/rt -- python3 $GRAFT_REPO_ROOT/tools/r04/round_timeline.py run float64 > $GRAFT_REPO_ROOT/$out/rt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/r04/tail_listing.py $out/rt 1.0 > $out/tail.txt 2>&1
rm -rf $out/rt
tail -70 $out/tail.txt
