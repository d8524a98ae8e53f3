#!/bin/bash
mkdir -p gpurun_out
{
echo "=== random tile test"; timeout 1200 python -m pytest tests/test_gpu_round5.py -q -k random_block 2>&1 | grep -E "^E|passed|failed" | head -20
} > gpurun_out/check_r5i.txt 2>&1
cat gpurun_out/check_r5i.txt | cut -c1-600
