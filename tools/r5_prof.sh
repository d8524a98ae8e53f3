#!/bin/bash
mkdir -p gpurun_out
{
echo "=== random tile test"; timeout 1200 python -m pytest tests/test_gpu_round5.py -q -k "random_block or tile" 2>&1 | grep -E "^E|passed|failed" | head -10
echo "=== prof round"; bash tools/prof_round.sh r05 2>&1 | tail -5
} > gpurun_out/check_r5j.txt 2>&1
cat gpurun_out/check_r5j.txt | cut -c1-400
