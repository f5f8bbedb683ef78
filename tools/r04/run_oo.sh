#!/bin/bash
# A/B on one box: stage loop of the f64 128 x 128 tile rotated by one k-step (libssa_rot.so) against the production tile
out=gpurun_out/r04oo; mkdir -p $out
R=$GRAFT_REPO_ROOT/superscreen_amd/lib/libssa_rot.so
for rep in 1 2; do
  timeout 300 python tools/probes/syrk_m_probe.py 8192 12288 16384 20224 > $out/m_base_$rep.txt 2>&1
  SSA_LIB_PATH=$R timeout 300 python tools/probes/syrk_m_probe.py 8192 12288 16384 20224 > $out/m_rot_$rep.txt 2>&1
done
for rep in 1 2; do
  timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 > $out/bench_base_$rep.json 2>$out/err
  SSA_LIB_PATH=$R timeout 600 python bench.py --no-extras --no-cpu-baseline --steps 10 --warmup 3 > $out/bench_rot_$rep.json 2>$out/err
done
grep -h "K= 512" $out/m_base_1.txt $out/m_rot_1.txt $out/m_base_2.txt $out/m_rot_2.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04oo/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],2), d["parity"]["max_rel_err_stream"] if "parity" in d else None)
PY
