#!/bin/bash
out=gpurun_out/r04ab; mkdir -p $out; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pt -- python3 $GRAFT_REPO_ROOT/tools/pass_trace.py > $GRAFT_REPO_ROOT/$out/pt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pass_trace.py analyse $out/pt > $out/pass_trace.txt 2>&1
rm -rf $out/pt
head -24 $out/pass_trace.txt
timeout 600 python -m pytest tests -m gpu -x -q --timeout 600 -k "biot or sheet or self_field or field_at or coupled or two_film_vs_oracle_medium" 2>&1 | tail -2
