#!/bin/bash
mkdir -p gpurun_out
SECONDS=0
timeout 1200 python bench.py > gpurun_out/bench_r5g.json 2> gpurun_out/bench_r5g.err
echo "bench wall seconds: $SECONDS"
tail -c 400 gpurun_out/bench_r5g.json
