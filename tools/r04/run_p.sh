#!/bin/bash
out=gpurun_out/r04p; mkdir -p $out
timeout 2400 python tools/collect_profiles.py r04 $out > $out/collect.log 2>&1; echo "collect rc=$?"
tail -3 $out/collect.log; ls $out
