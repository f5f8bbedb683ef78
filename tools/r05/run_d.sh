#!/bin/bash
# round 5, call D: after the barrier fix of tile_small_nt -- race hunt, whole suite, config-5 repeat, bench
out=gpurun_out/r05d; mkdir -p $out; rm -f $out/summary.txt
timeout 900 python -X faulthandler tools/chol_race_hunt.py 500 > $out/hunt_plain.txt 2>&1; echo "hunt rc=$? $(tail -1 $out/hunt_plain.txt)" >> $out/summary.txt
timeout 1500 python -X faulthandler -m pytest tests -q -m gpu --timeout 600 > $out/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $out/summary.txt; tail -3 $out/pytest_gpu.log >> $out/summary.txt
timeout 600 python -X faulthandler tools/config5_repeat.py 10 > $out/config5_repeat.txt 2>&1; echo "config5 repeat rc=$? $(tail -1 $out/config5_repeat.txt)" >> $out/summary.txt
timeout 900 python -X faulthandler bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" >> $out/summary.txt
cat $out/summary.txt
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05d/bench.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "roofline")})
    print("parity", {k: v for k, v in d.get("parity", {}).items() if k.startswith("max_rel")})
    e = d.get("extras", {})
    print({k: e[k] for k in e if "frac" in k or k.endswith("_ms")})
except Exception as exc:
    print("bench line unreadable:", exc)
PY
