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
import hashlib, os, subprocess, sys
sys.path.insert(0, os.getcwd())
from superscreen_amd import build
print("library_source_digest", build.source_digest())
print("library_is_current", build.is_current())
h = hashlib.sha256()
for root in ("superscreen_amd", "tests", "oracle", "include"):
    for d, _, files in sorted(os.walk(root)):
        if "__pycache__" in d or "/build" in d or d.endswith("/lib") or "_ref" in d:
            continue
        for f in sorted(files):
            if f.endswith((".py", ".hip", ".hpp", ".h", ".c", ".npz")):
                h.update(os.path.join(d, f).encode()); h.update(open(os.path.join(d, f), "rb").read())
for f in ("bench.py", "__graft_entry__.py"):
    h.update(f.encode()); h.update(open(f, "rb").read())
print("tree_digest", h.hexdigest())
PY
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
