#!/bin/bash
# Functional check of bench.py's N > 1 code path on a one-GPU box: two ranks share device 0 (CS_BENCH_SHARE_GPU=1: the collectives go
# through the torch.distributed callback over gloo, RCCL refuses two ranks on one device).  Second run: the watchdog of the multi-GPU BA
# leg, forced to fire.  Never a performance number.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
Q="--steps 4 --warmup 1 --steady-steps 0 --latency-calls 0 --lines-images 0 --rp-frames 0 --no-edge --no-measure-traffic --no-cpu-baseline --frames 200 --unique 20"
CS_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 $Q > gpurun_out/two_ranks.json 2> gpurun_out/two_ranks.err
echo "rc=$?"; tail -c 400 gpurun_out/two_ranks.err
CS_BENCH_SHARE_GPU=1 CS_BENCH_BA_LIMIT_S=0.2 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 $Q > gpurun_out/two_ranks_watchdog.json 2> gpurun_out/two_ranks_watchdog.err
echo "rc=$?"; tail -c 400 gpurun_out/two_ranks_watchdog.err
