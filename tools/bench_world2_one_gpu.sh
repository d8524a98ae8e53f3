#!/bin/bash
# Exercise bench.py's N > 1 code path (barriers, max-over-ranks timing, the sharded C5 leg, the all-gather leg) on a ONE-GPU box:
# two ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one device).  A code-path check, not a measurement.
export TSGU_BENCH_TEST_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 5 --warmup 3 --no-cpu-baseline "$@"
