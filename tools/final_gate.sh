#!/bin/bash
# Final gate of a round (run on the GPU box through gpurun, from the repo root):
#   the whole GPU suite exactly as the driver runs it (-x), smoke(), and a driver-style bench run,
#   with the digest of the sources the library was built from, so that the log can be matched to a commit:
#   `python tools/final_gate.py --check <log>` (CPU) compares that digest with the tree's.
# Usage: bash tools/final_gate.sh <tag>     -> gpurun_out/<tag>_gate/{gate.txt,pytest_gpu.log,smoke.log,bench.json}
tag=${1:-gate}; export GATE_TAG=$tag
out=gpurun_out/${tag}_gate; mkdir -p $out
log=$out/gate.txt; : > $log
echo "gate $(date -u +%FT%TZ) tag=$tag" >> $log
python - >> $log <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from superscreen_amd import build
print("library_source_digest", build.source_digest())
print("library_is_current", build.is_current())
PY
echo "tree_digest $(python tools/final_gate.py)" >> $log     # (the ONE implementation of the digest: tools/final_gate.py)
timeout 1800 python -X faulthandler -m pytest tests -x -q -m gpu --timeout 600 > $out/pytest_gpu.log 2>&1; echo "pytest_gpu_rc $?" >> $log; tail -1 $out/pytest_gpu.log >> $log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke_rc $?" >> $log; tail -1 $out/smoke.log >> $log
timeout 900 python -X faulthandler bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench_rc $?" >> $log
python - >> $log <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/%s_gate/bench.json" % __import__("os").environ.get("GATE_TAG", "gate")).read().strip().splitlines()[-1])
    p = d.get("parity", {})
    print("bench value", d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "roofline.frac", d["roofline"]["frac"],
          "parity stream", p.get("max_rel_err_stream"), "fluxoid", p.get("max_rel_err_fluxoid"))
except Exception as exc:
    print("bench line unreadable:", exc)
PY
cat $log
