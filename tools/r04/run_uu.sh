#!/bin/bash
out=gpurun_out/r04uu; mkdir -p $out
export BENCH_CONFIG2_K=40 BENCH_CONFIG3_K=30 BENCH_CONFIGH_ALT_K=40 BENCH_CONFIG4_K=30 BENCH_CONFIG5_K=30
# (1) one rank, small extras: the line with every extra
timeout 600 python bench.py --K 40 --steps 2 --warmup 1 --cpu-budget-s 5 > $out/n1.json 2> $out/n1.err; echo "n1 rc=$?"
# (2) two ranks on one GPU over gloo: collective extras complete
BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --K 40 --steps 2 --warmup 1 > $out/n2.json 2> $out/n2.err; echo "n2 rc=$?"
# (3) the same with a watchdog that fires at once: the line still goes out, marked
BENCH_SHARE_GPU=1 BENCH_COLLECTIVE_EXTRAS_TIMEOUT_S=0.05 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --K 40 --steps 2 --warmup 1 > $out/n2w.json 2> $out/n2w.err; echo "n2 watchdog rc=$?"
python - <<'PY'
import json
for f in ("n1","n2","n2w"):
    try:
        lines=[l for l in open(f"gpurun_out/r04uu/{f}.json").read().splitlines() if l.startswith("{")]
        d=json.loads(lines[-1]); e=d["extras"]
        print(f, len(lines), "line(s)", d["n_gpus"], round(d["value"],2), [k for k in e if k.endswith("_error")], "config5_solves_per_s" in e, "config4_strong_solves_per_s" in e, "cpu_baseline" in d)
    except Exception as ex:
        print(f, "FAILED", ex)
PY
tail -n 3 $out/n1.err $out/n2.err $out/n2w.err
