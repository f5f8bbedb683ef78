#!/bin/bash
out=gpurun_out/r04ac; mkdir -p $out; rm -f $out/summary.txt
timeout 1800 python -X faulthandler -m pytest tests -q -m gpu --timeout 600 > $out/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $out/summary.txt; tail -3 $out/pytest_gpu.log >> $out/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?" >> $out/summary.txt; tail -1 $out/smoke.log >> $out/summary.txt
timeout 2400 python tools/collect_profiles.py r04 $out > $out/collect.log 2>&1; echo "collect rc=$?" >> $out/summary.txt
timeout 900 python -X faulthandler bench.py --steps 20 --warmup 5 > $out/bench_driver_style.json 2> $out/bench.err; echo "bench rc=$?" >> $out/summary.txt
cat $out/summary.txt
