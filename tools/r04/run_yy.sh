#!/bin/bash
# A/B on one box: at most 256 / 240 / 224 / 192 workgroups per slice of the finishing passes beside the rounds
out=gpurun_out/r04yy; mkdir -p $out
L=$GRAFT_REPO_ROOT/superscreen_amd/lib
for rep in 1 2; do
  for w in base 240 224 192; do
    if [ $w = base ]; then unset SSA_LIB_PATH; else export SSA_LIB_PATH=$L/libssa_s$w.so; fi
    timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 > $out/bench_${w}_$rep.json 2>$out/err
  done
done
unset SSA_LIB_PATH
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04yy/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["ms_per_step"],2))
PY
