#!/bin/bash
# A/B on one box: triangular GEMV with the rows of a wave in pairs (g, n - 1 - g) against consecutive rows
out=gpurun_out/r04zz; mkdir -p $out
U=$GRAFT_REPO_ROOT/superscreen_amd/lib/libssa_unpaired.so
for rep in 1 2; do
  SSA_LIB_PATH=$U timeout 300 python tools/warm_solve_timing.py 2>/dev/null | tail -1
  timeout 300 python tools/warm_solve_timing.py 2>/dev/null | tail -1
done
for rep in 1 2; do
  SSA_LIB_PATH=$U timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 > $out/bench_unpaired_$rep.json 2>$out/err
  timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 > $out/bench_paired_$rep.json 2>$out/err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04zz/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["ms_per_step"],2))
PY
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 -k "chol or solve or headline or sweep" 2>&1 | tail -2
