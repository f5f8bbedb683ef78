#!/bin/bash
# the N = 8 control flow of bench.py on one GPU (gloo, small meshes): config 5 takes the helper-group placement
out=gpurun_out/r04bb; mkdir -p $out
export BENCH_SHARE_GPU=1 BENCH_CONFIG2_K=30 BENCH_CONFIG3_K=24 BENCH_CONFIGH_ALT_K=30 BENCH_CONFIG4_K=24 BENCH_CONFIG5_K=24
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 2 --warmup 1 --K 30 --no-cpu-baseline > $out/bench8.json 2> $out/bench8.err; echo "rc=$?"
tail -c 3000 $out/bench8.json; tail -5 $out/bench8.err
