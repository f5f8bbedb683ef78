#!/bin/bash
out=gpurun_out/r04kk; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 2 > $out/bench_traced.json 2> $out/bench_traced.err
timeout 600 python tools/r04/syrk_launch_table.py $out/trace > $out/syrk_launch_table.txt 2>&1
rm -rf $out/trace
cat $out/syrk_launch_table.txt | tail -50
