#!/bin/bash
# float32 with the diagonal blocks factored in float64: error attribution, float32 tests, bench with extras
out=gpurun_out/r04ff; mkdir -p $out
timeout 300 python tools/r04/f32_attrib.py 91 > $out/f32_attrib.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 -k "float32 or schedule_parts or chol" > $out/pytest_f32.txt 2>&1
timeout 900 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
cat $out/f32_attrib.txt; tail -5 $out/pytest_f32.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04ff/bench.json").read().strip().splitlines()[-1])
e=d["extras"]
print(d["value"], d["ms_per_step"], e.get("factorization_ms"))
for k,v in e.items():
    if "float32" in k or "f32" in k: print(k, v)
PY
